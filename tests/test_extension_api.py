"""The reference's extension-module API (north_star: "keeps the existing torch.autograd.Function / setup.py extension API"):
`_raymarching`, `_gridencoder`, `_freqencoder`, `_ffmlp` exist as importable torch-extension modules (focnerf_amd/csrc/ext, built by
__graft_entry__.build() / setup.py into focnerf_amd/ext) with the reference bindings' names and positional arguments.

CPU: the modules import, export every name the reference's bindings.cpp files define, and refuse CPU tensors with a RuntimeError; in
the build container the reference's OWN wrapper packages import on top of them with no edit (their `try: import _x as _backend` finds
these modules, so the JIT build of the CUDA sources behind the except-branch is never reached).
GPU: every shim entry point against focnerf_amd.backend (the ctypes route to the same C ABI), bit for bit."""
import os
import sys
import types

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXT = os.path.join(REPO, "focnerf_amd", "ext")
REF = "/root/reference"

# names defined by the reference's four bindings.cpp files
EXPECTED = {
    "_raymarching": ["packbits", "near_far_from_aabb", "sph_from_ray", "morton3D", "morton3D_invert", "march_rays_train", "composite_rays_train_forward",
                     "composite_rays_train_backward", "march_rays", "composite_rays"],
    "_gridencoder": ["grid_encode_forward", "grid_encode_backward", "grad_total_variation"],
    "_freqencoder": ["freq_encode_forward", "freq_encode_backward"],
    "_ffmlp": ["ffmlp_forward", "ffmlp_inference", "ffmlp_backward", "allocate_splitk", "free_splitk"],
}


def _mods():
    if EXT not in sys.path:
        sys.path.insert(0, EXT)
    import importlib
    return {n: importlib.import_module(n) for n in EXPECTED}


def test_modules_import_and_export_the_reference_names():
    mods = _mods()
    for name, fns in EXPECTED.items():
        assert os.path.dirname(mods[name].__file__) == EXT
        for fn in fns:
            assert callable(getattr(mods[name], fn)), f"{name}.{fn}"


def test_cpu_tensors_raise_like_torch_check():
    mods = _mods()
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        mods["_freqencoder"].freq_encode_forward(torch.zeros(4, 3), 4, 3, 4, 27, torch.zeros(4, 27))
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        mods["_gridencoder"].grid_encode_forward(torch.zeros(4, 3), torch.zeros(8, 2), torch.zeros(2, dtype=torch.int32), torch.zeros(1, 4, 2), 4, 3, 2, 1, 0.5, 16, None, 0,
                                                 False, 0)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        h = torch.zeros(128, 32, dtype=torch.half)
        mods["_ffmlp"].ffmlp_forward(h, h, 128, 32, 16, 64, 2, 0, 6, h, h)
    mods["_ffmlp"].allocate_splitk(8)
    mods["_ffmlp"].free_splitk()
    with pytest.raises(TypeError):                                   # pybind11 argument checking: wrong arity is a TypeError, as with the reference module
        mods["_raymarching"].packbits(torch.zeros(8))
    # _raymarching: the reference's functions check nothing; a CPU tensor must not reach a kernel as a bad pointer
    rm = mods["_raymarching"]
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        rm.near_far_from_aabb(torch.zeros(4, 3), torch.zeros(4, 3), torch.zeros(6), 4, 0.2, torch.zeros(4), torch.zeros(4))
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        rm.packbits(torch.zeros(64), 8, 0.5, torch.zeros(8, dtype=torch.uint8))
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        rm.composite_rays_train_forward(torch.zeros(8), torch.zeros(8, 3), torch.zeros(8, 2), torch.zeros(2, 3, dtype=torch.int32), 8, 2, 1e-4, torch.zeros(2), torch.zeros(2),
                                        torch.zeros(2, 3))
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        rm.morton3D(torch.zeros(4, 3, dtype=torch.int32), 4, torch.zeros(4, dtype=torch.int32))


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_reference_wrapper_packages_run_unedited_on_the_shims():
    """/root/reference/{raymarching,gridencoder,freqencoder,ffmlp} imported as they are: their loader takes the installed-extension
    branch. (`turtle` is stubbed: ffmlp/ffmlp.py:2 has a stray import of it that needs tkinter)"""
    mods = _mods()                                                   # must be importable BEFORE the reference packages are touched
    sys.dont_write_bytecode = True
    saved_path, saved_mods = list(sys.path), dict(sys.modules)
    try:
        for k in [k for k in sys.modules if k.split(".")[0] in ("raymarching", "gridencoder", "freqencoder", "ffmlp")]:
            del sys.modules[k]
        sys.modules.setdefault("turtle", types.SimpleNamespace(backward=None, forward=None))
        sys.path.insert(0, REF)
        sys.path.insert(0, EXT)
        import raymarching as r
        import gridencoder as g
        import freqencoder as f
        import ffmlp as m
        for pkg, inner, ext in ((r, "raymarching", "_raymarching"), (g, "grid", "_gridencoder"), (f, "freq", "_freqencoder"), (m, "ffmlp", "_ffmlp")):
            assert pkg.__file__.startswith(REF)
            backend = sys.modules[f"{pkg.__name__}.{inner}"]._backend
            assert backend is mods[ext], f"{pkg.__name__} did not pick up the installed extension"
            assert f"{pkg.__name__}.backend" not in sys.modules       # the JIT-building fallback module was never imported
        enc = g.GridEncoder(desired_resolution=2048)
        assert enc.embeddings.shape[0] == 6119864 and enc.output_dim == 32
        mlp = m.FFMLP(32, 16, 64, 2)
        assert mlp.weights.numel() == 64 * (32 + 64 + 16)
        fe = f.FreqEncoder(input_dim=3, degree=4)
        assert fe.output_dim == 27
    finally:
        sys.path[:] = saved_path
        for k in list(sys.modules):
            if k not in saved_mods:
                del sys.modules[k]


# ------------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_shims_match_the_ctypes_backend_bit_for_bit():
    import numpy as np
    from focnerf_amd import backend, synthetic
    from focnerf_amd.gridencoder import GridEncoder
    mods = _mods()
    dev = "cuda"
    torch.manual_seed(0)
    # ---- _raymarching: near/far, march, composite forward + backward
    bound = 2.0
    grid = synthetic.analytic_density_grid(2, device=dev)
    bits = torch.empty(grid.numel() // 8, dtype=torch.uint8, device=dev)
    mods["_raymarching"].packbits(grid.contiguous(), grid.numel() // 8, 0.01, bits)
    bits2 = torch.empty_like(bits)
    backend._raymarching.packbits(grid.contiguous(), grid.numel() // 8, 0.01, bits2)
    assert torch.equal(bits, bits2)
    ro, rd = synthetic.make_view_rays(32, 32, 2, 1, seed=1, device=dev)
    o, d = ro[0].contiguous(), rd[0].contiguous()
    N = o.shape[0]
    aabb = torch.tensor([-2.0, -2, -2, 2, 2, 2], device=dev)
    res = []
    for be in (mods["_raymarching"], backend._raymarching):
        nears, fars = torch.empty(N, device=dev), torch.empty(N, device=dev)
        be.near_far_from_aabb(o, d, aabb, N, 0.2, nears, fars)
        M = N * 64
        xyzs, dirs, deltas = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
        rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
        counter = torch.zeros(2, dtype=torch.int32, device=dev)
        be.march_rays_train(o, d, bits, bound, 1 / 128, 1024, N, 2, 128, M, nears, fars, xyzs, dirs, deltas, rays, counter, torch.zeros(N, device=dev))
        g = torch.Generator(device=dev).manual_seed(5)
        sig = torch.rand(M, device=dev, generator=g) * 5
        rgb = torch.rand(M, 3, device=dev, generator=g)
        ws, dep, img = torch.empty(N, device=dev), torch.empty(N, device=dev), torch.empty(N, 3, device=dev)
        be.composite_rays_train_forward(sig, rgb, deltas, rays, M, N, 1e-4, ws, dep, img)
        gs, gc = torch.zeros(M, device=dev), torch.zeros(M, 3, device=dev)
        be.composite_rays_train_backward(torch.ones(N, device=dev), torch.ones(N, 3, device=dev), sig, rgb, deltas, rays, ws, img, M, N, 1e-4, gs, gc)
        res.append((nears, fars, xyzs, deltas, rays, counter, ws, dep, img, gs, gc))
    assert int(res[0][5][0]) > 0
    for a, b in zip(*res):
        assert torch.equal(a, b)
    # ---- _gridencoder: forward [L,B,C] fp16 + fp32 with dy_dx, backward (binned path for the NeRF table, atomic for dy_dx)
    enc = GridEncoder(desired_resolution=2048).cuda()
    enc.embeddings.data.uniform_(-1, 1)
    B, L = 5000, 16
    x = torch.rand(B, 3, device=dev)
    S = float(np.log2(enc.per_level_scale))
    for dt in (torch.half, torch.float32):
        emb = enc.embeddings.detach().to(dt).contiguous()
        outs, gembs = [], []
        for be in (mods["_gridencoder"], backend._gridencoder):
            out = torch.empty(L, B, 2, device=dev, dtype=dt)
            be.grid_encode_forward(x, emb, enc.offsets, out, B, 3, 2, L, S, 16, None, 0, False, 0)
            grad = (torch.randn(L, B, 2, device=dev, generator=torch.Generator(device=dev).manual_seed(9)) * 0.01).to(dt)
            gemb = torch.zeros_like(emb)
            be.grid_encode_backward(grad, x, emb, enc.offsets, gemb, B, 3, 2, L, S, 16, None, None, 0, False, 0)
            outs.append(out); gembs.append(gemb)
        assert torch.equal(outs[0], outs[1]) and outs[0].abs().max() > 0
        assert torch.equal(gembs[0], gembs[1]) and gembs[0].abs().max() > 0       # the binned backward is deterministic
    # ---- _freqencoder
    outs = []
    for be in (mods["_freqencoder"], backend._freqencoder):
        out = torch.empty(B, 27, device=dev)
        be.freq_encode_forward(x, B, 3, 4, 27, out)
        gi = torch.zeros(B, 3, device=dev)
        be.freq_encode_backward(torch.ones(B, 27, device=dev), out, B, 3, 4, 27, gi)
        outs.append((out, gi))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # ---- _ffmlp: training forward (stores activations), inference, backward
    Bp = 1024
    w = (torch.randn(64 * (32 + 64 + 16), device=dev) * 0.2).half()
    xin = (torch.randn(Bp, 32, device=dev) * 0.5).half()
    gy = (torch.randn(Bp, 16, device=dev) * 0.1).half()
    res = []
    for be in (mods["_ffmlp"], backend._ffmlp):
        fb = torch.empty(2, Bp, 64, dtype=torch.half, device=dev)
        y = torch.empty(Bp, 16, dtype=torch.half, device=dev)
        be.ffmlp_forward(xin, w, Bp, 32, 16, 64, 2, 0, 6, fb, y)
        yi = torch.empty(Bp, 16, dtype=torch.half, device=dev)
        be.ffmlp_inference(xin, w, Bp, 32, 16, 64, 2, 0, 6, torch.empty(Bp, 64, dtype=torch.half, device=dev), yi)
        bb = torch.zeros(2, Bp, 64, dtype=torch.half, device=dev)
        gx = torch.zeros(Bp, 32, dtype=torch.half, device=dev)
        gw = torch.zeros_like(w)
        be.ffmlp_backward(gy, xin, w, fb, Bp, 32, 16, 64, 2, 0, 6, True, bb, gx, gw)
        res.append((y, yi, fb, gx, bb, gw))
    for k, (a, b) in enumerate(zip(*res)):
        if k < 5:
            assert torch.equal(a, b), k
        else:                                                                     # dW: fp32 atomics across workgroups
            torch.testing.assert_close(a.float(), b.float(), rtol=2e-2, atol=2e-3)
    with pytest.raises(RuntimeError, match="hidden_dim"):
        mods["_ffmlp"].ffmlp_forward(xin, w, Bp, 32, 16, 48, 2, 0, 6, fb, y)          # the library's message comes through as a RuntimeError


@pytest.mark.gpu
def test_raymarching_shim_rejects_wrong_dtype_and_layout():
    """The reference's _raymarching checks nothing and dispatches on the scalar type; its wrappers only ever hand it contiguous fp32. The
    shim refuses anything else with a RuntimeError instead of reinterpreting half / double bits as fp32 or walking a strided tensor."""
    rm = _mods()["_raymarching"]
    N = 64
    o = torch.zeros(N, 3, device="cuda")
    d = torch.nn.functional.normalize(torch.ones(N, 3, device="cuda"), dim=-1)
    aabb = torch.tensor([-1., -1., -1., 1., 1., 1.], device="cuda")
    nears, fars = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    rm.near_far_from_aabb(o, d, aabb, N, 0.2, nears, fars)                                  # the valid call
    with pytest.raises(RuntimeError, match="float32"):
        rm.near_far_from_aabb(o.half(), d, aabb, N, 0.2, nears, fars)
    with pytest.raises(RuntimeError, match="float32"):
        rm.near_far_from_aabb(o, d.double(), aabb, N, 0.2, nears, fars)
    with pytest.raises(RuntimeError, match="contiguous"):
        rm.near_far_from_aabb(torch.zeros(3, N, device="cuda").t(), d, aabb, N, 0.2, nears, fars)
    with pytest.raises(RuntimeError, match="int tensor"):
        rm.morton3D(torch.zeros(N, 3, device="cuda"), N, torch.zeros(N, dtype=torch.int32, device="cuda"))
    with pytest.raises(RuntimeError, match="uint8"):
        rm.packbits(torch.zeros(64, device="cuda"), 8, 0.5, torch.zeros(8, dtype=torch.int32, device="cuda"))
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        rm.near_far_from_aabb(o.cpu(), d, aabb, N, 0.2, nears, fars)
