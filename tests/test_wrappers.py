"""The Python operator wrappers against tests/golden/wrappers.npz — the reference's own gridencoder/grid.py and ffmlp/ffmlp.py driven on
the CPU with oracle-backed stub backends (tests/golden/make_golden.py). CPU part: constructor arithmetic (level offsets, scale, weight
blob, seed-42 init). GPU part: the same calls through focnerf_amd on the device reproduce the wrapper-level results."""
import os

import numpy as np
import pytest
import torch

from util import to_np, assert_half_close

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(os.path.dirname(HERE), "focnerf_amd", "libfocnerf_hip.so")
needs_lib = pytest.mark.skipif(not os.path.exists(LIB), reason="libfocnerf_hip.so not built")
SMALL = dict(input_dim=3, num_levels=8, level_dim=2, base_resolution=4, log2_hashmap_size=12, desired_resolution=96)


def _fx():
    return np.load(os.path.join(HERE, "golden", "wrappers.npz"))


@needs_lib
def test_constructors_match_reference_wrappers():
    from focnerf_amd.gridencoder import GridEncoder
    from focnerf_amd.ffmlp import FFMLP
    g = _fx()
    big = GridEncoder(desired_resolution=2048)
    assert np.array_equal(big.offsets.numpy(), g["big_offsets"]) and big.per_level_scale == float(g["big_per_level_scale"])
    assert big.embeddings.shape == (int(g["big_offsets"][-1]), 2) and big.output_dim == 32
    small = GridEncoder(**SMALL)
    assert np.array_equal(small.offsets.numpy(), g["ge_offsets"]) and small.per_level_scale == float(g["ge_per_level_scale"])
    assert small.output_dim == int(g["ge_output_dim"])
    mlp = FFMLP(input_dim=32, output_dim=3, hidden_dim=64, num_layers=2)
    assert np.array_equal(mlp.weights.detach().numpy(), g["ff_weights"]), "weight blob size / seed-42 initialisation (ffmlp.py:120-144)"
    assert list(g["ff_allocate_splitk"]) == [mlp.num_layers + 1]
    # the reference pads every batch to the next multiple of 128 — a full extra block when it is already aligned (ffmlp.py:157-159)
    assert list(g["ff_inference_B"]) == [256] and list(g["ff_forward_B"]) == [256] and int(g["ff_inference_B_aligned"]) == 256


@pytest.mark.gpu
def test_grid_encoder_module_matches_reference_wrapper():
    from focnerf_amd.gridencoder import GridEncoder
    g = _fx()
    enc = GridEncoder(**SMALL).cuda()
    enc.embeddings.data.copy_(torch.from_numpy(g["ge_embeddings"]))
    x = torch.from_numpy(g["ge_x"]).cuda().requires_grad_(True)
    y = enc(x, bound=2)                                             # fp32, no autocast: what the CPU run of the reference computed
    assert y.shape == g["ge_y"].shape and y.dtype == torch.float32
    assert np.array_equal(to_np(y), g["ge_y"]), "forward through normalisation, kernel and layout"
    y.backward(torch.from_numpy(g["ge_gy"]).cuda())
    np.testing.assert_allclose(to_np(enc.embeddings.grad), g["ge_grad_embeddings"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(to_np(x.grad), g["ge_grad_x"], atol=1e-3, rtol=1e-3)


@pytest.mark.gpu
def test_ffmlp_module_matches_reference_wrapper():
    from focnerf_amd.ffmlp import FFMLP
    g = _fx()
    mlp = FFMLP(input_dim=32, output_dim=3, hidden_dim=64, num_layers=2).cuda()
    assert np.array_equal(to_np(mlp.weights), g["ff_weights"])
    x = torch.from_numpy(g["ff_x"]).cuda()
    mlp.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        y = mlp(x)
    assert y.shape == (200, 3) and y.dtype == torch.float16
    # the oracle behind the fixture accumulates in fp16 like the reference's WMMA fragments, the MFMA kernels in fp32: a few half-ulps
    assert_half_close(to_np(y), g["ff_y_eval"], ulps=8, atol=4e-3, what="inference")
    mlp.train()
    xt = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.float16):
        yt = mlp(xt)
    assert_half_close(to_np(yt), g["ff_y_train"], ulps=8, atol=4e-3, what="training forward")
    yt.backward(torch.from_numpy(g["ff_gy"]).cuda())
    assert_half_close(to_np(xt.grad), g["ff_grad_x"], ulps=8, atol=2e-3, what="grad_inputs")
    gw, gw_ref = to_np(mlp.weights.grad).astype(np.float32), g["ff_grad_w"].astype(np.float32)
    assert gw.shape == gw_ref.shape
    assert np.abs(gw - gw_ref).max() <= 2e-2 * np.abs(gw_ref).max() + 1e-3


@pytest.mark.gpu
def test_raymarching_wrappers_match_reference_wrappers():
    """tests/golden/raymarching_wrappers.npz: the reference's raymarching.py wrappers on an oracle-backed `_raymarching` stub. Same calls
    through focnerf_amd.raymarching on the device: shapes after the mean_count / align / force_all_rays sizing and slicing, sample
    positions and ray tables bit for bit, composites within 1e-4 (`__expf`)."""
    from focnerf_amd import raymarching as rm
    g = np.load(os.path.join(HERE, "golden", "raymarching_wrappers.npz"))
    H, C, bound = int(g["H"]), int(g["C"]), float(g["bound"])
    dev = "cuda"
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    idx = torch.arange(H ** 3, dtype=torch.int32, device=dev)
    coords = rm.morton3D_invert(idx)
    assert torch.equal(rm.morton3D(coords), idx) and bool(g["morton_roundtrip_ok"])
    bitfield = rm.packbits(t("grid"), 0.5)
    assert np.array_equal(to_np(bitfield), g["bitfield"])
    o, d, aabb = t("rays_o"), t("rays_d"), t("aabb")
    nears, fars = rm.near_far_from_aabb(o, d, aabb, 0.2)
    assert np.array_equal(to_np(nears), g["nears"]) and np.array_equal(to_np(fars), g["fars"])
    cases = (("first", dict(mean_count=-1, align=128)), ("steady", dict(mean_count=700, align=128)), ("force", dict(mean_count=700, align=128, force_all_rays=True)),
             ("noalign", dict(mean_count=-1, align=-1)))
    for name, kw in cases:
        counter = torch.zeros(2, dtype=torch.int32, device=dev)
        xyzs, dirs, deltas, rays = rm.march_rays_train(o, d, bound, bitfield, C, H, nears, fars, counter, kw.get("mean_count", -1), False, kw.get("align", -1),
                                                       kw.get("force_all_rays", False), 1 / 128, 256)
        assert xyzs.shape == g[f"mt_{name}_xyzs"].shape, name
        assert np.array_equal(to_np(counter), g[f"mt_{name}_counter"]), name
        assert np.array_equal(to_np(rays), g[f"mt_{name}_rays"]), name
        for a, k in ((xyzs, "xyzs"), (dirs, "dirs"), (deltas, "deltas")):
            assert np.array_equal(to_np(a).view(np.uint32), g[f"mt_{name}_{k}"].view(np.uint32)), (name, k)
    # composite_rays_train, forward and through autograd
    sig, rgb = t("ct_sigmas").requires_grad_(True), t("ct_rgbs").requires_grad_(True)
    ws, dep, img = rm.composite_rays_train(sig, rgb, t("mt_first_deltas"), t("mt_first_rays"), 1e-4)
    np.testing.assert_allclose(to_np(ws), g["ct_ws"], atol=1e-4)
    np.testing.assert_allclose(to_np(dep), g["ct_depth"], atol=1e-4)
    np.testing.assert_allclose(to_np(img), g["ct_image"], atol=1e-4)
    (ws * t("ct_gws")).sum().add((img * t("ct_gimg")).sum()).backward()
    np.testing.assert_allclose(to_np(sig.grad), g["ct_grad_sigmas"], atol=2e-4, rtol=1e-3)
    np.testing.assert_allclose(to_np(rgb.grad), g["ct_grad_rgbs"], atol=1e-4)
    # one inference iteration
    N = o.shape[0]
    rays_alive = torch.arange(N, dtype=torch.int32, device=dev)
    rays_t = nears.clone()
    x2, d2, dl2 = rm.march_rays(N, 3, rays_alive, rays_t, o, d, bound, bitfield, C, H, nears, fars, 128, False, 1 / 128, 256)
    assert x2.shape == g["mi_xyzs"].shape
    assert np.array_equal(to_np(x2).view(np.uint32), g["mi_xyzs"].view(np.uint32)) and np.array_equal(to_np(dl2).view(np.uint32), g["mi_deltas"].view(np.uint32))
    wsum, dpt, im = torch.zeros(N, device=dev), torch.zeros(N, device=dev), torch.zeros(N, 3, device=dev)
    rm.composite_rays(N, 3, rays_alive, rays_t, t("mi_sigmas"), t("mi_rgbs"), dl2, wsum, dpt, im, 1e-2)
    assert np.array_equal(to_np(rays_alive), g["mi_rays_alive"])
    np.testing.assert_allclose(to_np(rays_t), g["mi_rays_t"], atol=1e-6)
    np.testing.assert_allclose(to_np(wsum), g["mi_ws"], atol=1e-4)
    np.testing.assert_allclose(to_np(dpt), g["mi_depth"], atol=1e-4)
    np.testing.assert_allclose(to_np(im), g["mi_image"], atol=1e-4)


@pytest.mark.gpu
def test_run_cuda_matches_reference_renderer():
    """The reference's NeRFRenderer.run_cuda (nerf/renderer.py:243-352) was run on top of its own raymarching wrappers (oracle backends)
    with an analytic field — two training calls (first epochs; a too small mean_count that drops rays) and the inference loop.
    focnerf_amd.renderer.NeRFRenderer.run_cuda with the same field on the GPU must give the same images, depths and step counters."""
    from focnerf_amd.renderer import NeRFRenderer
    from test_oracle import _golden_sigma
    g = np.load(os.path.join(HERE, "golden", "raymarching_wrappers.npz"))
    H, bound = int(g["H"]), float(g["bound"])

    class ToyField(NeRFRenderer):
        def forward(self, x, d):
            c = torch.tensor([0.1, -0.05, 0.2], device=x.device)
            r2 = ((x - c) * (x - c)).sum(-1)
            a = torch.clamp(1.0 - r2 * 2.5, min=0.0)
            q = ((x + 0.4) * (x + 0.4)).sum(-1)
            b = torch.clamp(1.0 - q * 16.0, min=0.0)
            return a * a * 40.0 + b * 3.0, 0.5 + 0.25 * (x[..., :3] * 0.5 + d)

    m = ToyField(bound=bound, cuda_ray=True).cuda()
    m.grid_size = H
    m.density_grid = torch.from_numpy(g["grid"]).cuda()
    m.density_bitfield = torch.from_numpy(g["bitfield"]).cuda()
    o, d = torch.from_numpy(g["rc_rays_o"]).cuda(), torch.from_numpy(g["rc_rays_d"]).cuda()
    m.train()
    tr0 = m.run_cuda(o[None], d[None], dt_gamma=1 / 128, bg_color=None, perturb=False, force_all_rays=False, max_steps=256, T_thresh=1e-4)
    m.mean_count = 640
    tr1 = m.run_cuda(o[None], d[None], dt_gamma=1 / 128, bg_color=0.25, perturb=False, force_all_rays=False, max_steps=256, T_thresh=1e-4)
    m.eval()
    with torch.no_grad():
        ev = m.run_cuda(o[None], d[None], dt_gamma=1 / 128, bg_color=None, perturb=False, max_steps=256, T_thresh=1e-4, device_compaction=False)
        ev2 = m.run_cuda(o[None], d[None], dt_gamma=1 / 128, bg_color=None, perturb=False, max_steps=256, T_thresh=1e-4, device_compaction=True)
    assert np.array_equal(to_np(m.step_counter), g["rc_step_counter"])
    for got, img, dep in ((tr0, "rc_train0_image", "rc_train0_depth"), (tr1, "rc_train1_image", "rc_train1_depth"), (ev, "rc_eval_image", "rc_eval_depth"),
                          (ev2, "rc_eval_image", "rc_eval_depth")):
        np.testing.assert_allclose(to_np(got["image"][0]), g[img], atol=1e-4, err_msg=img)
        ok = np.isfinite(g[dep])
        assert np.array_equal(np.isfinite(to_np(got["depth"][0])), ok), dep
        np.testing.assert_allclose(to_np(got["depth"][0])[ok], g[dep][ok], atol=1e-4, err_msg=dep)
    np.testing.assert_allclose(to_np(tr0["weights_sum"]), g["rc_train0_ws"], atol=1e-4)
    assert (g["rc_train1_image"] != g["rc_train0_image"]).any(), "the second call used another background and dropped rays"
