"""GPU parity: fully fused MLP (MFMA kernels) vs the CPU oracle.

Tolerance model (SURVEY.md §7 "fp16-accumulate vs fp32-accumulate"): every layer output is rounded to
fp16 once; the kernel accumulates in fp32 on the matrix cores in MFMA order, the oracle (acc_mode=0)
accumulates in fp32 in k order, so results agree up to fp32 summation order BEFORE the fp16 rounding:
at most 1 half-ulp per layer, compounding over layers -> a few half-ulps + a small absolute term.
With small-integer data every product and sum is exact and the comparison is bit-exact: that is the
test that pins the MFMA operand / accumulator lane maps and the chained-operand k permutation."""
import math

import numpy as np
import pytest
import torch

import oracle
from util import to_np, assert_half_close

pytestmark = pytest.mark.gpu


def _be():
    from focnerf_amd.backend import _ffmlp
    return _ffmlp


def _n_params(I, Hd, nl):
    return Hd * (I + Hd * (nl - 1) + 16)


def _run_forward(x, W, I, Hd, nl, act, train):
    be = _be()
    B = x.shape[0]
    xt, Wt = torch.from_numpy(x).cuda(), torch.from_numpy(W).cuda()
    out = torch.empty(B, 16, dtype=torch.float16, device="cuda")
    if train:
        fb = torch.empty(nl, B, Hd, dtype=torch.float16, device="cuda")
        be.ffmlp_forward(xt, Wt, B, I, 16, Hd, nl, act, 6, fb, out)
        return to_np(out), to_np(fb)
    # hidden 256 runs layer by layer, in place in the [B, hidden] buffer the reference allocates for every width (ffmlp.py:41)
    scratch = torch.empty(B, Hd, dtype=torch.float16, device="cuda") if Hd > 128 else None
    be.ffmlp_inference(xt, Wt, B, I, 16, Hd, nl, act, 6, scratch, out)
    return to_np(out)


SHAPES = [(32, 64, 2), (32, 64, 3), (16, 16, 2), (48, 32, 4), (64, 128, 2), (128, 128, 3), (16, 64, 5),
          (32, 256, 2), (256, 256, 3), (48, 256, 4),          # hidden 256 (ffmlp.cu:652-658): csrc/ffmlp_wide.hip
          (192, 64, 2), (256, 128, 2), (144, 32, 3)]          # input layers wider than 128 at hidden <= 128 (the reference's dynamic input layer, ffmlp.cu:151-239)


@pytest.mark.parametrize("I,Hd,nl", SHAPES)
@pytest.mark.parametrize("act", [0, 6])
def test_forward_exact_on_integer_data(I, Hd, nl, act):
    """Small integers: all partial sums are exact in fp32, so every summation order rounds to the same fp16 -> bit-exact."""
    rng = np.random.default_rng(I * 1000 + Hd + nl)
    B = 256
    # sparse small weights keep |activations| < 2048 so fp16 represents every integer exactly
    W = np.zeros(_n_params(I, Hd, nl), np.float16)
    nz = rng.random(W.size) < (4.0 / max(I, Hd))
    W[nz] = rng.integers(-2, 3, nz.sum()).astype(np.float16)
    x = rng.integers(-3, 4, (B, I)).astype(np.float16)
    ref_out, ref_fb = oracle.ffmlp_forward(x, W, I, Hd, nl, act)
    # integer sums stay far below 2^24 (exact in fp32) and below the fp16 overflow; values above 2048 round to
    # the same fp16 on both sides because the exact sum is identical
    assert np.abs(ref_fb.astype(np.float32)).max() < 30000 and np.abs(ref_out.astype(np.float32)).max() < 30000
    assert np.count_nonzero(ref_out) > 100, "degenerate test data"
    out, fb = _run_forward(x, W, I, Hd, nl, act, True)
    assert np.array_equal(fb, ref_fb), "forward_buffer differs: MFMA operand/accumulator map is wrong"
    assert np.array_equal(out, ref_out)
    assert np.array_equal(_run_forward(x, W, I, Hd, nl, act, False), ref_out)


@pytest.mark.parametrize("I,Hd,nl", SHAPES)
def test_forward_random(I, Hd, nl):
    rng = np.random.default_rng(7 + I + Hd + nl)
    B = 512
    W = (rng.uniform(-1, 1, _n_params(I, Hd, nl)) * math.sqrt(3 / Hd)).astype(np.float16)
    x = rng.standard_normal((B, I)).astype(np.float16)
    ref_out, ref_fb = oracle.ffmlp_forward(x, W, I, Hd, nl, 0)
    out, fb = _run_forward(x, W, I, Hd, nl, 0, True)
    assert_half_close(fb[0], ref_fb[0], ulps=1.0, atol=1e-6, what="layer 0")
    assert_half_close(fb, ref_fb, ulps=2.0 * nl, atol=2e-3, what="forward_buffer")
    assert_half_close(out, ref_out, ulps=2.0 * (nl + 1), atol=4e-3, what="outputs")
    # against the WMMA-like half-accumulate model of the reference: same numbers up to fp16 accumulation error
    ref_h = oracle.ffmlp_forward(x, W, I, Hd, nl, 0, training=False, acc_mode=1)
    # measured on MI355X over the thirteen shapes (tools/measure_ffmlp_literal.py, gpurun_out/r5k_ffmlp_literal.txt): worst |difference| / max(1, |output|max)
    # 1.74e-3 (256 -> 256 x 3), 5.3e-4 / 5.7e-4 on the two NeRF shapes; 4e-3 pins it with a margin of 2.3 (round 4 asserted 5e-2). The network-level
    # figure is tests/test_gpu_network.py::test_end_to_end_distance_to_reference_literal_numerics
    assert np.abs(out.astype(np.float32) - ref_h.astype(np.float32)).max() < 4e-3 * max(1.0, np.abs(ref_h.astype(np.float32)).max())
    assert np.array_equal(_run_forward(x, W, I, Hd, nl, 0, False), out), "inference and training kernels must agree bit for bit"


@pytest.mark.parametrize("I,Hd,nl", SHAPES)
@pytest.mark.parametrize("act", [0, 6])
def test_backward_exact_on_integer_data(I, Hd, nl, act):
    be = _be()
    rng = np.random.default_rng(I * 77 + Hd + nl)
    B = 256
    W = np.zeros(_n_params(I, Hd, nl), np.float16)
    nz = rng.random(W.size) < (3.0 / max(I, Hd))
    W[nz] = rng.integers(-2, 3, nz.sum()).astype(np.float16)
    x = rng.integers(-2, 3, (B, I)).astype(np.float16)
    g = np.zeros((B, 16), np.float16)
    gz = rng.random(g.shape) < 0.25
    g[gz] = rng.integers(-2, 3, gz.sum()).astype(np.float16)
    ref_out, ref_fb = oracle.ffmlp_forward(x, W, I, Hd, nl, act)
    gw_r, gi_r, bb_r = oracle.ffmlp_backward(g, x, W, ref_fb, I, Hd, nl, act, True)
    assert np.abs(bb_r.astype(np.float32)).max() < 30000 and np.abs(gw_r.astype(np.float32)).max() < 30000
    t = lambda a: torch.from_numpy(a).cuda()
    bb = torch.empty(nl, B, Hd, dtype=torch.float16, device="cuda")
    gi = torch.empty(B, I, dtype=torch.float16, device="cuda")
    gw = torch.empty(W.size, dtype=torch.float16, device="cuda")
    be.ffmlp_backward(t(g), t(x), t(W), t(ref_fb), B, I, 16, Hd, nl, act, 6, True, bb, gi, gw)
    assert np.array_equal(to_np(bb), bb_r), "backward_buffer differs"
    assert np.array_equal(to_np(gi), gi_r), "grad_inputs differs"
    assert np.array_equal(to_np(gw), gw_r), "grad_weights differs"
    # without grad_inputs the weight gradients are unchanged
    gw2 = torch.empty_like(gw)
    be.ffmlp_backward(t(g), t(x), t(W), t(ref_fb), B, I, 16, Hd, nl, act, 6, False, bb, torch.zeros(1, dtype=torch.float16, device="cuda"), gw2)
    assert torch.equal(gw, gw2)


@pytest.mark.parametrize("I,Hd,nl", [(32, 64, 2), (32, 64, 3), (48, 32, 4), (64, 128, 2), (32, 256, 3), (192, 64, 2), (256, 128, 2)])
def test_backward_random(I, Hd, nl):
    be = _be()
    rng = np.random.default_rng(99 + I + Hd + nl)
    B = 1024
    W = (rng.uniform(-1, 1, _n_params(I, Hd, nl)) * math.sqrt(3 / Hd)).astype(np.float16)
    x = rng.standard_normal((B, I)).astype(np.float16)
    g = (rng.standard_normal((B, 16)) * 0.05).astype(np.float16)
    out, fb = _run_forward(x, W, I, Hd, nl, 0, True)
    gw_r, gi_r, bb_r = oracle.ffmlp_backward(g, x, W, fb, I, Hd, nl, 0, True)
    t = lambda a: torch.from_numpy(a).cuda()
    bb = torch.empty(nl, B, Hd, dtype=torch.float16, device="cuda")
    gi = torch.empty(B, I, dtype=torch.float16, device="cuda")
    gw = torch.empty(W.size, dtype=torch.float16, device="cuda")
    be.ffmlp_backward(t(g), t(x), t(W), t(fb), B, I, 16, Hd, nl, 0, 6, True, bb, gi, gw)
    assert_half_close(to_np(bb), bb_r, ulps=2.0 * nl, atol=2e-4, what="backward_buffer")
    assert_half_close(to_np(gi), gi_r, ulps=2.0 * (nl + 1), atol=5e-4, what="grad_inputs")
    assert_half_close(to_np(gw), gw_r, ulps=4.0, atol=2e-3, what="grad_weights")


def test_module_padding_autograd_and_init():
    """FFMLP module: seed-42 init, batch padding to a multiple of 128 (+128 when already aligned), autocast, autograd."""
    from focnerf_amd.ffmlp import FFMLP
    net = FFMLP(32, 3, 64, 3).cuda()
    assert net.weights.numel() == 64 * (32 + 64 * 2 + 16)
    torch.manual_seed(42)
    want = torch.zeros(net.weights.numel()).uniform_(-math.sqrt(3 / 64), math.sqrt(3 / 64))
    assert torch.equal(net.weights.detach().cpu(), want), "seed-42 init of ffmlp.py:141-144"
    for B in (1, 127, 128, 300):
        x = torch.randn(B, 32, device="cuda", requires_grad=True)
        net.train()
        with torch.autocast("cuda", dtype=torch.float16):
            y = net(x)
        assert y.shape == (B, 3) and y.dtype == torch.float16
        ref = oracle.ffmlp_forward(to_np(x.detach().half()), to_np(net.weights.detach().half()), 32, 64, 3, 0, training=False)
        assert_half_close(to_np(y), ref[:, :3], ulps=8, atol=4e-3, what=f"module forward B={B}")
        net.zero_grad()
        y.float().pow(2).sum().backward()
        assert net.weights.grad is not None and net.weights.grad.dtype == torch.float32 and torch.isfinite(net.weights.grad).all()
        assert x.grad is not None and x.grad.shape == (B, 32)
        net.eval()
        with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
            y2 = net(x)
        assert torch.equal(y2, y.detach())
    # fp32 inputs outside autocast are rejected like CHECK_IS_HALF does
    with pytest.raises(RuntimeError):
        net(torch.randn(128, 32, device="cuda"))


def test_full_batch_rows_are_independent():
    """BASELINE size: B = 2 097 152 rows (4096 rays x 512 samples). Rows are independent, so a random subset
    evaluated by the oracle must equal the same rows of the full launch; the weight gradient of the full
    batch must equal the sum of the gradients of its two halves (linearity over the batch)."""
    be = _be()
    rng = np.random.default_rng(5)
    I, Hd, nl = 32, 64, 2
    B = 4096 * 512
    W = (rng.uniform(-1, 1, _n_params(I, Hd, nl)) * math.sqrt(3 / Hd)).astype(np.float16)
    Wt = torch.from_numpy(W).cuda()
    x = torch.randn(B, I, device="cuda").half()
    out = torch.empty(B, 16, dtype=torch.float16, device="cuda")
    fb = torch.empty(nl, B, Hd, dtype=torch.float16, device="cuda")
    be.ffmlp_forward(x, Wt, B, I, 16, Hd, nl, 0, 6, fb, out)
    sel = torch.from_numpy(rng.choice(B, 2048, replace=False)).cuda()
    sel = torch.cat([sel, torch.tensor([0, 63, 64, B - 1], device="cuda")])
    ref_out, ref_fb = oracle.ffmlp_forward(to_np(x[sel]), W, I, Hd, nl, 0)
    assert_half_close(to_np(out[sel]), ref_out, ulps=6, atol=4e-3, what="subset rows")
    assert_half_close(to_np(fb[:, sel]), ref_fb, ulps=4, atol=2e-3, what="subset forward_buffer")
    g = (torch.randn(B, 16, device="cuda") * 0.01).half()

    def bwd(lo, hi):
        n = hi - lo
        bb = torch.empty(nl, n, Hd, dtype=torch.float16, device="cuda")
        gi = torch.empty(n, I, dtype=torch.float16, device="cuda")
        gw = torch.empty(W.size, dtype=torch.float16, device="cuda")
        be.ffmlp_backward(g[lo:hi].contiguous(), x[lo:hi].contiguous(), Wt, fb[:, lo:hi].contiguous(), n, I, 16, Hd, nl, 0, 6, True, bb, gi, gw)
        return gw.float(), gi
    gw_full, gi_full = bwd(0, B)
    gw_a, gi_a = bwd(0, B // 2)
    gw_b, gi_b = bwd(B // 2, B)
    assert torch.equal(gi_full[: B // 2], gi_a) and torch.equal(gi_full[B // 2:], gi_b)
    scale = gw_full.abs().max()
    assert (gw_full - (gw_a + gw_b)).abs().max() <= 4e-3 * scale + 1e-3


@pytest.mark.parametrize("B", [1, 63, 65, 333])
def test_ragged_batch_matches_oracle(B):
    """Batches that are not a multiple of the 64-row tile: the last tile is computed on clamped rows and never stored;
    weight gradients must not see the phantom rows."""
    be = _be()
    rng = np.random.default_rng(B)
    I, Hd, nl = 32, 64, 3
    W = (rng.uniform(-1, 1, _n_params(I, Hd, nl)) * math.sqrt(3 / Hd)).astype(np.float16)
    x = rng.standard_normal((B, I)).astype(np.float16)
    g = (rng.standard_normal((B, 16)) * 0.05).astype(np.float16)
    out, fb = _run_forward(x, W, I, Hd, nl, 0, True)
    ref_out, ref_fb = oracle.ffmlp_forward(x, W, I, Hd, nl, 0)
    assert_half_close(out, ref_out, ulps=8, atol=4e-3, what="outputs")
    assert_half_close(fb, ref_fb, ulps=6, atol=2e-3, what="forward_buffer")
    gw_r, gi_r, bb_r = oracle.ffmlp_backward(g, x, W, fb, I, Hd, nl, 0, True)
    t = lambda a: torch.from_numpy(a).cuda()
    # guard regions after every buffer: nothing may be written past row B
    bb = torch.full((nl, B + 64, Hd), 7.0, dtype=torch.float16, device="cuda")
    gi = torch.full((B + 64, I), 7.0, dtype=torch.float16, device="cuda")
    gw = torch.empty(W.size, dtype=torch.float16, device="cuda")
    bbv = torch.empty(nl, B, Hd, dtype=torch.float16, device="cuda")
    be.ffmlp_backward(t(g), t(x), t(W), t(fb), B, I, 16, Hd, nl, 0, 6, True, bbv, gi[:B], gw)
    assert torch.all(gi[B:] == 7.0), "grad_inputs written past row B"
    assert_half_close(to_np(bbv), bb_r, ulps=6, atol=2e-4, what="backward_buffer")
    assert_half_close(to_np(gi[:B]), gi_r, ulps=8, atol=5e-4, what="grad_inputs")
    assert_half_close(to_np(gw), gw_r, ulps=4.0, atol=2e-3, what="grad_weights")


@pytest.mark.parametrize("I,Hd,nl", [(32, 64, 2), (32, 64, 3), (48, 32, 4), (16, 16, 2), (64, 64, 4)])
@pytest.mark.parametrize("B", [1000, 128 * 300 + 17])
@pytest.mark.parametrize("act", [0, 6])
def test_backward_recompute_is_bit_identical(I, Hd, nl, B, act):
    """forward_buffer = NULL at the C ABI: the forward stores no activations and the fused backward re-evaluates them from
    the inputs. Same MFMA sequence as the forward kernel -> activations, masks, backward_buffer, grad_inputs are the same bits;
    grad_weights are fp32 sums of the same products (atomics across workgroups: order may differ in the last fp32 bit)."""
    be = _be()
    rng = np.random.default_rng(I + Hd + nl + B)
    W = (rng.uniform(-1, 1, _n_params(I, Hd, nl)) * math.sqrt(3 / Hd)).astype(np.float16)
    x = rng.standard_normal((B, I)).astype(np.float16)
    g = (rng.standard_normal((B, 16)) * 0.05).astype(np.float16)
    t = lambda a: torch.from_numpy(a).cuda()
    xt, Wt, gt = t(x), t(W), t(g)
    out_s = torch.empty(B, 16, dtype=torch.float16, device="cuda")
    fb = torch.empty(nl, B, Hd, dtype=torch.float16, device="cuda")
    be.ffmlp_forward(xt, Wt, B, I, 16, Hd, nl, act, 6, fb, out_s)
    out_r = torch.empty_like(out_s)
    be.ffmlp_forward(xt, Wt, B, I, 16, Hd, nl, act, 6, None, out_r)
    assert torch.equal(out_s, out_r)
    res = {}
    for name, buf in (("stored", fb), ("recompute", None)):
        bb = torch.full((nl, B, Hd), 3.0, dtype=torch.float16, device="cuda")
        gi = torch.full((B + 8, I), 7.0, dtype=torch.float16, device="cuda")
        gw = torch.empty(W.size, dtype=torch.float16, device="cuda")
        be.ffmlp_backward(gt, xt, Wt, buf, B, I, 16, Hd, nl, act, 6, True, bb, gi[:B], gw)
        assert torch.all(gi[B:] == 7.0)
        res[name] = (bb, gi[:B].clone(), gw)
    assert torch.equal(res["stored"][0], res["recompute"][0]), "backward_buffer"
    assert torch.equal(res["stored"][1], res["recompute"][1]), "grad_inputs"
    # fp32 atomics from ~1000 workgroups in arbitrary order: absolute error scales with the magnitude of the partial sums
    gw_tol = 5e-4 * max(1.0, float(res["stored"][2].float().abs().max()))
    assert_half_close(to_np(res["recompute"][2]), to_np(res["stored"][2]), ulps=2.0, atol=gw_tol, what="grad_weights")
    # and without backward_buffer / grad_inputs
    gw2 = torch.empty(W.size, dtype=torch.float16, device="cuda")
    be.ffmlp_backward(gt, xt, Wt, None, B, I, 16, Hd, nl, act, 6, False, None, torch.zeros(1, dtype=torch.float16, device="cuda"), gw2)
    assert_half_close(to_np(gw2), to_np(res["stored"][2]), ulps=2.0, atol=gw_tol, what="grad_weights (no dx)")


def test_module_recompute_env(monkeypatch):
    """FFMLP autograd through both storage policies gives the same gradients."""
    from focnerf_amd.ffmlp import FFMLP
    grads = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FOC_MLP_RECOMPUTE", mode)
        net = FFMLP(32, 3, 64, 3).cuda().train()
        torch.manual_seed(5)
        x = torch.randn(777, 32, device="cuda", dtype=torch.float16, requires_grad=True)
        with torch.autocast("cuda", dtype=torch.float16):
            y = net(x)
        (y.float() ** 2).sum().backward()
        grads[mode] = (x.grad.clone(), net.weights.grad.clone(), y.detach().clone())
    assert torch.equal(grads["1"][2], grads["0"][2])
    assert torch.equal(grads["1"][0], grads["0"][0])
    assert torch.allclose(grads["1"][1], grads["0"][1], rtol=2e-3, atol=1e-4)


def test_hidden_256_module_trains_and_infers():
    """FFMLP(hidden_dim=256) through autograd: stored activations (the layer-by-layer path has no re-evaluating form), ragged batch, inference
    equal to the training forward, null-buffer calls rejected with the library's message."""
    from focnerf_amd.ffmlp import FFMLP
    net = FFMLP(32, 3, 256, 2).cuda()
    assert net.weights.numel() == 256 * (32 + 256 + 16)
    x = torch.randn(333, 32, device="cuda", requires_grad=True)
    net.train()
    with torch.autocast("cuda", dtype=torch.float16):
        y = net(x)
    ref = oracle.ffmlp_forward(to_np(x.detach().half()), to_np(net.weights.detach().half()), 32, 256, 2, 0, training=False)
    assert_half_close(to_np(y), ref[:, :3], ulps=8, atol=4e-3, what="hidden 256 module forward")
    y.float().pow(2).sum().backward()
    assert torch.isfinite(net.weights.grad).all() and net.weights.grad.abs().sum() > 0 and x.grad.shape == (333, 32)
    net.eval()
    with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
        assert torch.equal(net(x), y.detach())
    be = _be()
    xh, w = x.detach().half(), net.weights.detach().half()
    with pytest.raises(RuntimeError, match="inference_buffer"):
        be.ffmlp_inference(xh, w, 333, 32, 16, 256, 2, 0, 6, None, torch.empty(333, 16, dtype=torch.float16, device="cuda"))


@pytest.mark.parametrize("I,Hd", [(192, 64), (256, 128)])
def test_wide_input_module_trains_and_infers(I, Hd):
    """FFMLP with an input layer wider than 128 at hidden <= 128 (ffmlp.cu:151-239 takes any 16 m that fits shared memory; mlp_check refused it until
    round 5): forward against the oracle, autograd through the stored-activation backward (split-K weight gradients over 24 / 32 input tiles),
    inference equal to the training forward, ragged batch; 272 inputs are still refused with the library's message."""
    from focnerf_amd.ffmlp import FFMLP
    net = FFMLP(I, 3, Hd, 2).cuda()
    x = torch.randn(777, I, device="cuda", requires_grad=True)
    net.train()
    with torch.autocast("cuda", dtype=torch.float16):
        y = net(x)
    ref = oracle.ffmlp_forward(to_np(x.detach().half()), to_np(net.weights.detach().half()), I, Hd, 2, 0, training=False)
    assert_half_close(to_np(y), ref[:, :3], ulps=8, atol=4e-3, what="wide-input module forward")
    y.float().pow(2).sum().backward()
    assert torch.isfinite(net.weights.grad).all() and net.weights.grad.abs().sum() > 0 and x.grad.shape == (777, I) and torch.isfinite(x.grad).all()
    # the weight gradient of the input layer against the float64 outer-product sum of the oracle's deltas
    gw_r, gi_r, _ = oracle.ffmlp_backward((2 * ref.astype(np.float32) * (np.arange(16) < 3)).astype(np.float16), to_np(x.detach().half()),
                                          to_np(net.weights.detach().half()), oracle.ffmlp_forward(to_np(x.detach().half()), to_np(net.weights.detach().half()), I, Hd, 2, 0)[1],
                                          I, Hd, 2, 0, True)
    scale = float(np.abs(gw_r.astype(np.float32)).max())
    assert np.abs(to_np(net.weights.grad).astype(np.float32) - gw_r.astype(np.float32)).max() <= 2e-2 * scale
    net.eval()
    with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
        assert torch.equal(net(x), y.detach())
    be = _be()
    with pytest.raises(RuntimeError, match="up to 256"):
        be.ffmlp_inference(torch.zeros(128, 272, dtype=torch.float16, device="cuda"), torch.zeros(Hd * (272 + Hd + 16), dtype=torch.float16, device="cuda"),
                           128, 272, 16, Hd, 2, 0, 6, None, torch.empty(128, 16, dtype=torch.float16, device="cuda"))


ACT_NAMES = {1: "exponential", 2: "sine", 3: "sigmoid", 4: "squareplus", 5: "softplus"}


@pytest.mark.parametrize("I,Hd,nl", [(32, 64, 2), (48, 32, 3), (64, 128, 2), (32, 256, 2)])
@pytest.mark.parametrize("act", [1, 2, 3, 4, 5])
def test_other_hidden_activations_forward_and_backward(I, Hd, nl, act):
    """The reference's hidden activations beyond ReLU / None (ffmlp/src/utils.h:424-589; ffmlp.py:87-95 codes 1..5): forward, stored
    activations, activation gradients, weight gradients and input gradients against the oracle's restatement — the activation in fp32 on the
    half-rounded sum, the backward factor from the stored post-activation with the reference's half arithmetic, Sine's backward a
    pass-through exactly as the reference leaves it. Device expf / sinf / logf differ from libm in the last fp32 bits: after the rounding to
    half nearly every value is the oracle's, the rest one or two half-ulps away."""
    rng = np.random.default_rng(100 * act + Hd)
    B = 777
    x = (rng.standard_normal((B, I)) * 0.5).astype(np.float16)
    W = (rng.standard_normal(_n_params(I, Hd, nl)) * (0.6 / math.sqrt(Hd))).astype(np.float16)
    out, fb = _run_forward(x, W, I, Hd, nl, act, True)
    out_ref, fb_ref = oracle.ffmlp_forward(x, W, I, Hd, nl, act)
    assert np.isfinite(out_ref.astype(np.float32)).all() and np.abs(fb_ref.astype(np.float32)).max() > 0.1
    assert_half_close(fb, fb_ref, ulps=3.0, atol=2e-3, what=f"{ACT_NAMES[act]}: stored activations")
    assert_half_close(out, out_ref, ulps=3.0, atol=4e-3, what=f"{ACT_NAMES[act]}: outputs")
    assert (fb[0] == fb_ref[0]).mean() > 0.9, f"first layer: only {(fb[0] == fb_ref[0]).mean():.3f} of the activations are the oracle's bits"
    out_inf = _run_forward(x, W, I, Hd, nl, act, False)
    assert np.array_equal(out_inf, out), "inference and training forward are the same arithmetic"
    # ---- backward on the ORACLE's forward buffer (so that both sides transfer through identical post-activations)
    be = _be()
    g = (rng.standard_normal((B, 16)) * 0.1).astype(np.float16)
    gw_ref, gi_ref, bb_ref = oracle.ffmlp_backward(g, x, W, fb_ref, I, Hd, nl, act, True)
    gt, xt, Wt, fbt = (torch.from_numpy(a).cuda() for a in (g, x, W, fb_ref))
    bb = torch.zeros(nl, B, Hd, dtype=torch.float16, device="cuda")
    gi = torch.zeros(B, I, dtype=torch.float16, device="cuda")
    gw = torch.zeros_like(Wt)
    be.ffmlp_backward(gt, xt, Wt, fbt, B, I, 16, Hd, nl, act, 6, True, bb, gi, gw)
    s = np.abs(bb_ref.astype(np.float32)).max()
    assert s > 0 and np.abs(to_np(bb).astype(np.float32) - bb_ref.astype(np.float32)).max() <= 4e-3 * s
    assert (to_np(bb)[0] == bb_ref[0]).mean() > 0.97            # the first transfer sees identical operands: the half arithmetic is the reference's
    for name, a, b in (("grad_inputs", to_np(gi), gi_ref), ("grad_weights", to_np(gw), gw_ref)):
        s = np.abs(b.astype(np.float32)).max()
        assert s > 0 and np.abs(a.astype(np.float32) - b.astype(np.float32)).max() <= 5e-3 * s, name
    if act == 2:                                                # Sine: the reference's backward leaves the fragment untouched (utils.h:549-553)
        Wo = W[Hd * I + (nl - 1) * Hd * Hd:].reshape(16, Hd).astype(np.float32)
        plain = (g.astype(np.float32) @ Wo).astype(np.float16)
        assert_half_close(to_np(bb)[0], plain, ulps=2.0, atol=1e-4, what="sine: gradient passes through unchanged")


def test_other_activations_through_the_module_and_autograd():
    """FFMLP(activation='sigmoid' | ...) trains through autograd: non-ReLU activations keep the reference's data flow (stored activations +
    gradient buffer) instead of the single-pass backward, and the fused paths that assume ReLU stay away from them."""
    from focnerf_amd.ffmlp import FFMLP, single_pass_backward
    from focnerf_amd.field import field_fusable
    from focnerf_amd.gridencoder import GridEncoder
    assert single_pass_backward(32, 64, 2, 0) and not single_pass_backward(32, 64, 2, 3)
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2).cuda()
    for name in ("sigmoid", "softplus", "squareplus", "exponential", "sine"):
        net = FFMLP(32, 3, 64, 2, activation=name).cuda().train()
        assert net.activation in ACT_NAMES and not field_fusable(enc, net)
        x = (torch.randn(500, 32, device="cuda") * 0.5).half().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.float16):
            y = net(x)
        (y.float() ** 2).sum().backward()
        ref, _ = oracle.ffmlp_forward(to_np(x), to_np(net.weights).astype(np.float16), 32, 64, 2, net.activation)
        assert_half_close(to_np(y), ref[:, :3], ulps=3.0, atol=4e-3, what=name)
        assert torch.isfinite(net.weights.grad).all() and net.weights.grad.abs().max() > 0 and x.grad.abs().max() > 0
