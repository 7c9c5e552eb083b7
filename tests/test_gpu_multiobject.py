"""GPU parity: MONeRFNetwork (focnerf_amd/multiobject.py; reference nerf/multiobjectnetwork.py). `foc_mo_select` against the oracle, bit for
bit, on ties / signed zeros / infinities / NaN and ragged sizes; the class against the reference's loop written with the reference's torch
ops (stack, max, take_along_dim) over the same K object networks loaded from per-object checkpoints."""
import numpy as np
import pytest
import torch

import oracle
from test_multiobject import cases, reference_step
from util import to_np

pytestmark = pytest.mark.gpu


def _bits(a):
    return a.view(np.uint16 if a.dtype == np.float16 else np.uint32)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("width", [1, 3, 15, 16, 64])
@pytest.mark.parametrize("n_random", [0, 1, 27, 28, 29, 1000, 100003])          # + 100 special pairs: 127 / 128 / 129 samples = a full and a ragged last wave
def test_mo_select_equals_the_oracle_bitwise(dtype, width, n_random):
    from focnerf_amd.combine import HipCombineOps
    a, fa, b, fb = cases(dtype, n_random, width, 11 + width)
    want_s, want_f = oracle.mo_select(a.numpy(), fa.numpy(), b.numpy(), fb.numpy())
    bs, bf = b.cuda().clone(), fb.cuda().clone()
    HipCombineOps.mo_select(a.cuda(), fa.cuda(), bs, bf)
    got_s, got_f = to_np(bs), to_np(bf)
    nan = np.isnan(want_s)
    assert np.array_equal(np.isnan(got_s), nan)
    assert np.array_equal(_bits(got_s)[~nan], _bits(np.ascontiguousarray(want_s))[~nan])
    assert np.array_equal(_bits(got_f), _bits(np.ascontiguousarray(want_f)))
    # and torch's own ops on the device (what the reference would run there)
    ts, tf = reference_step(a.cuda(), fa.cuda(), b.cuda(), fb.cuda())
    assert torch.equal(torch.isnan(ts), torch.isnan(bs)) and torch.equal(ts[~torch.isnan(ts)], bs[~torch.isnan(bs)]) and torch.equal(tf, bf)


def test_mo_select_refuses_what_it_cannot_serve():
    from focnerf_amd.combine import HipCombineOps
    s, f = torch.zeros(8, device="cuda"), torch.zeros(8, 3, device="cuda")
    with pytest.raises(RuntimeError):
        HipCombineOps.mo_select(s, f.half(), s.clone(), f.clone())                   # mixed element types
    with pytest.raises(RuntimeError):
        HipCombineOps.mo_select(s, f[:, :2], s.clone(), f.clone())                   # row widths differ (and a non-contiguous view)
    with pytest.raises(RuntimeError):
        HipCombineOps.mo_select(s.double(), f.double(), s.double(), f.double())
    with pytest.raises(RuntimeError):
        HipCombineOps.mo_select(s, torch.zeros(8, 65, device="cuda"), s.clone(), torch.zeros(8, 65, device="cuda"))     # > 64 elements per row
    HipCombineOps.mo_select(s[:0], f[:0], s[:0].clone(), f[:0].clone())              # empty: nothing to do, no launch


def _objects(tmp_path, K):
    """K object networks with distinct weights, saved in the reference trainer's checkpoint layout."""
    from focnerf_amd.network_foc import NeRFNetwork
    paths, models = [], []
    for k in range(K):
        m = NeRFNetwork(bound=1).cuda().eval()
        g = torch.Generator(device="cuda").manual_seed(50 + k)      # (after construction: FFMLP's init re-seeds the global generator, ffmlp.py:141)
        with torch.no_grad():
            m.encoder.embeddings.uniform_(-1.0, 1.0, generator=g)
            for net in (m.sigma_net, m.color_net):
                net.weights.mul_(1.0 + 0.2 * torch.randn(net.weights.shape, generator=g, device="cuda"))
        p = tmp_path / f"obj{k}.pth"
        torch.save({"epoch": 1, "global_step": 10, "model": {n: t.detach().cpu() for n, t in m.state_dict().items()}}, str(p))
        paths.append(str(p))
        models.append(m)
    return paths, models


def test_monerf_density_and_color_equal_the_reference_loop(tmp_path, fp16=True):
    """multiobjectnetwork.py:44-97 with its own torch ops over the K loaded networks against MONeRFNetwork: same sigma, geo_feat and colour,
    bit for bit; the object networks come from the checkpoints (the reference's get_model_with_checkpoint drops what it read: see the module
    docstring) and are loaded once."""
    from focnerf_amd.multiobject import MONeRFNetwork
    from focnerf_amd.network_foc import NeRFNetwork
    K, n = 3, 5000
    paths, models = _objects(tmp_path, K)
    g = torch.Generator().manual_seed(4)
    x = (torch.rand(n, 3, generator=g) * 2 - 1).cuda()
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).cuda()
    yolo = (None, None, torch.randn(144, generator=g).cuda())
    mo = MONeRFNetwork(paths, NeRFNetwork, fp16=fp16, nw_kwargs={"bound": 1})
    mo.to(torch.device("cuda"))
    out = mo.density(x)
    assert len(mo.objects()) == K and mo.objects() is mo.objects()                  # resident: built once
    for a, b in zip(mo.objects(), models):
        assert torch.equal(a.encoder.embeddings, b.encoder.embeddings)              # and they ARE the checkpoints' weights

    def loop(density_only, **ckw):
        best_s = best_r = None
        with torch.no_grad():
            for m in models:
                with torch.autocast("cuda", dtype=torch.float16, enabled=fp16):
                    dens = m.density(x)
                    rows = dens['geo_feat'] if density_only else m.color(x, d, yolo, **ckw)
                if best_s is None:
                    best_s, best_r = dens['sigma'], rows
                else:
                    best_s, idx = torch.max(torch.stack([dens['sigma'], best_s]), dim=0, keepdim=True)
                    best_s = best_s.squeeze(0)
                    best_r = torch.take_along_dim(torch.stack([rows, best_r]), idx.unsqueeze(-1), dim=0).squeeze(0)
        return best_s, best_r

    want_s, want_g = loop(True)
    assert out['sigma'].dtype == want_s.dtype and torch.equal(out['sigma'], want_s) and torch.equal(out['geo_feat'], want_g)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        assert len(torch.unique(torch.stack([m.density(x)['sigma'] for m in models]).argmax(0))) == K  # every object wins somewhere
    # color(): the renderer's call shape (nerf/renderer.py:187-188: mask of the samples worth a colour, the merged geo_feat)
    mask = torch.rand(n, generator=g).cuda() < 0.6
    want_c = loop(False, mask=mask, geo_feat=out['geo_feat'])[1]
    got_c = mo.color(x, d, yolo, mask=mask, geo_feat=out['geo_feat'])
    assert got_c.dtype == want_c.dtype and torch.equal(got_c, want_c)
    assert float(got_c[~mask].abs().max()) == 0.0 and float(got_c[mask].abs().max()) > 0.0
    want_c2 = loop(False, geo_feat=out['geo_feat'])[1]
    assert torch.equal(mo.color(x, d, yolo, geo_feat=out['geo_feat']), want_c2)


def test_monerf_identical_objects_and_one_object(tmp_path):
    """All ties (K copies of one object) and K = 1: the answer is that object's own."""
    from focnerf_amd.multiobject import MONeRFNetwork
    paths, models = _objects(tmp_path, 1)
    x = (torch.rand(777, 3) * 2 - 1).cuda()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        own = models[0].density(x)
    for ck in ([paths[0]], [paths[0]] * 3):
        mo = MONeRFNetwork(ck, fp16=True, nw_kwargs={"bound": 1})
        mo.to(torch.device("cuda"))
        out = mo.density(x)
        assert torch.equal(out['sigma'], own['sigma']) and torch.equal(out['geo_feat'], own['geo_feat'])
    with pytest.raises(RuntimeError):
        MONeRFNetwork([], nw_kwargs={"bound": 1}).to(torch.device("cuda")).density(x)
    # fp16=False is autocast(enabled=False) as in the reference: an FFMLP network then meets fp32 inputs and says so (CHECK_IS_HALF, ffmlp.cu:638)
    with pytest.raises(RuntimeError, match="half"):
        MONeRFNetwork([paths[0]], fp16=False, nw_kwargs={"bound": 1}).to(torch.device("cuda")).density(x)
