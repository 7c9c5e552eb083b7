"""CPU: the oracle's restatement of MONeRFNetwork's running select (nerf/multiobjectnetwork.py:66-82) against the torch ops the reference
calls — torch.max over stack([new, best]) with take_along_dim — on values chosen to hit every branch of the rule: ties (the new object
wins), +-0, +-inf, NaN on either side and on both."""
import numpy as np
import pytest
import torch

import oracle

SPECIALS = [0.0, -0.0, 1.0, -1.0, float("inf"), float("-inf"), float("nan"), 65504.0, 6e-8, 3.5]


def reference_step(sigma_new, feat_new, sigma_best, feat_best):
    """The body of the reference's loop for one further object (multiobjectnetwork.py:66-82), its ops verbatim in meaning."""
    best, idx = torch.max(torch.stack([sigma_new, sigma_best]), dim=0, keepdim=True)
    feat = torch.take_along_dim(torch.stack([feat_new, feat_best]), idx.unsqueeze(-1), dim=0)
    return best.squeeze(0), feat.squeeze(0)


def cases(dtype, n_random, width, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.tensor([x for x in SPECIALS for _ in SPECIALS], dtype=torch.float32)
    b = torch.tensor([y for _ in SPECIALS for y in SPECIALS], dtype=torch.float32)
    ra = torch.randn(n_random, generator=g).abs()
    rb = torch.where(torch.rand(n_random, generator=g) < 0.3, ra, torch.randn(n_random, generator=g).abs())     # 30 % exact ties
    a, b = torch.cat([a, ra]).to(dtype), torch.cat([b, rb]).to(dtype)
    fa = torch.randn(a.numel(), width, generator=g).to(dtype)
    fb = torch.randn(a.numel(), width, generator=g).to(dtype)
    return a, fa, b, fb


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("width", [3, 15])
def test_oracle_select_is_torch_max_and_take_along_dim(dtype, width):
    a, fa, b, fb = cases(dtype, 5000, width, 3)
    want_s, want_f = reference_step(a, fa, b, fb)
    got_s, got_f = oracle.mo_select(a.numpy(), fa.numpy(), b.numpy(), fb.numpy())
    bits = np.uint16 if dtype == torch.float16 else np.uint32
    nan = torch.isnan(want_s).numpy()
    assert np.array_equal(np.isnan(got_s), nan)                                    # NaN where torch says NaN (payloads are free)
    assert np.array_equal(got_s[~nan].view(bits), want_s.numpy()[~nan].view(bits))   # -0 / +0 kept apart
    assert np.array_equal(got_f.view(bits), want_f.numpy().view(bits))


def test_three_objects_in_sequence_later_checkpoint_takes_ties():
    s = [torch.tensor([1.0, 2.0, 2.0, 0.0]), torch.tensor([1.0, 1.0, 3.0, 0.0]), torch.tensor([0.5, 2.0, 3.0, 0.0])]
    f = [torch.full((4, 3), float(k)) for k in range(3)]
    bs, bf = s[0], f[0]
    os_, of_ = s[0].numpy(), f[0].numpy()
    for k in (1, 2):
        bs, bf = reference_step(s[k], f[k], bs, bf)
        os_, of_ = oracle.mo_select(s[k].numpy(), f[k].numpy(), os_, of_)
    assert bf[:, 0].tolist() == [1.0, 2.0, 2.0, 2.0]            # ties go to the LATEST object that reaches the maximum
    assert np.array_equal(of_, bf.numpy()) and np.array_equal(os_, bs.numpy())
