"""GPU parity: gridencoder / freqencoder (Python operator API -> C ABI -> HIP) vs the CPU oracle.
Forward: the kernels evaluate the same fp32 fmaf sequence as oracle acc_mode=1, so fp32 AND fp16 outputs
are bit-exact against it; against the reference-literal half accumulation (acc_mode=0) fp16 outputs are
within 1 half-ulp. Backward sums are atomics (order dependent, like the reference): tolerance only."""
import numpy as np
import pytest
import torch

import oracle
from util import to_np, assert_half_close, assert_bits_equal

pytestmark = pytest.mark.gpu


def _setup(D, C, L, H, log2_hash, desired, seed, dtype, gridtype="hash", align_corners=False):
    from focnerf_amd.gridencoder import level_offsets
    pls = np.exp2(np.log2(desired / H) / (L - 1))
    off = level_offsets(D, L, pls, H, log2_hash, align_corners)
    rng = np.random.default_rng(seed)
    table = rng.uniform(-1, 1, (int(off[-1]), C)).astype(dtype)
    return pls, float(np.log2(pls)), off, table


def _points(B, D, seed, oob=True):
    rng = np.random.default_rng(seed)
    x = rng.random((B, D)).astype(np.float32)
    x[0] = 0.0
    x[1] = 1.0
    if oob and B > 8:
        x[2, 0] = -0.01
        x[3, D - 1] = 1.0001
        x[4] = np.float32(1.0) - np.float32(1e-7)
    return x


def _be():
    from focnerf_amd.backend import _gridencoder
    return _gridencoder


CASES = [
    # D, C, L, H, log2_hash, desired, gridtype, align, interp
    (3, 2, 16, 16, 19, 2048, 0, False, 0),      # the NeRF default (bound 1)
    (3, 2, 16, 16, 19, 4096, 0, False, 0),      # bound 2
    (3, 1, 8, 16, 15, 512, 0, False, 0),
    (3, 4, 8, 16, 15, 512, 1, False, 0),        # tiled
    (3, 8, 4, 8, 14, 128, 0, True, 1),          # align_corners + smoothstep
    (2, 2, 12, 16, 17, 2048, 0, False, 1),
    # D = 4, 5 (gridencoder.cu:393-398: the reference dispatches D in {2,3,4,5}): runtime-D kernels, csrc/gridencoder_nd.hip
    (4, 2, 8, 8, 14, 64, 0, False, 0),
    (4, 4, 5, 4, 12, 20, 1, True, 1),           # tiled, align_corners, smoothstep
    (5, 2, 6, 4, 13, 24, 0, False, 0),
    (5, 1, 4, 4, 12, 12, 0, False, 1),
]
CASES_BWD = CASES[:5] + CASES[6:]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("dtype", [np.float32, np.float16])
def test_forward_bit_exact(case, dtype):
    D, C, L, H, lh, desired, gridtype, ac, interp = case
    pls, S, off, table = _setup(D, C, L, H, lh, desired, 1, dtype, align_corners=ac)
    B = 5000
    x = _points(B, D, 2)
    ref, ref_dy = oracle.grid_encode_forward(x, table, off, D, C, L, S, H, True, gridtype, ac, interp, acc_mode=1)
    tdt = torch.float32 if dtype == np.float32 else torch.float16
    xt, tt, ot = torch.from_numpy(x).cuda(), torch.from_numpy(table).cuda(), torch.from_numpy(off).cuda()
    be = _be()
    # reference layout [L,B,C]
    out = torch.empty(L, B, C, dtype=tdt, device="cuda")
    dy = torch.empty(B, L * D * C, dtype=tdt, device="cuda")
    be.grid_encode_forward(xt, tt, ot, out, B, D, C, L, S, H, dy, gridtype, ac, interp)
    assert_bits_equal(to_np(out), ref, "outputs [L,B,C]")
    got_dy = to_np(dy).reshape(B, L, D, C)
    if dtype == np.float32:
        assert np.array_equal(got_dy, ref_dy)
    else:
        assert_half_close(got_dy, ref_dy, ulps=1.0, what="dy_dx")
    # fused layout [B, L*C]
    out2 = torch.empty(B, L * C, dtype=tdt, device="cuda")
    be.grid_encode_forward(xt, tt, ot, out2, B, D, C, L, S, H, None, gridtype, ac, interp, out_bl=True)
    assert np.array_equal(to_np(out2), np.transpose(ref, (1, 0, 2)).reshape(B, L * C))
    # out-of-range points encode to exact zeros on every level (gridencoder.cu:119-135)
    assert np.all(to_np(out)[:, 2] == 0) and np.all(to_np(out)[:, 3] == 0)
    if dtype == np.float16:
        lit = oracle.grid_encode_forward(x, table, off, D, C, L, S, H, False, gridtype, ac, interp, acc_mode=0)
        # the reference rounds the running sum to half after each of the 2^D corners: up to ~2^D/2 half-ulps of the
        # largest partial sum (|table| <= 1 here, ulp 2^-11..2^-10)
        assert_half_close(to_np(out), lit, ulps=1.0, atol=(1 << D) * 0.5 * 2.0 ** -10, what="vs reference-literal half accumulation")


@pytest.mark.parametrize("case", CASES_BWD)
@pytest.mark.parametrize("dtype", [np.float32, np.float16])
@pytest.mark.parametrize("atomic", ["0", "1"])      # 0: partition + LDS accumulation (D=3,C=2), 1: scattered-atomic kernel
def test_backward(case, dtype, atomic, monkeypatch):
    monkeypatch.setenv("FOCNERF_GRID_ATOMIC", atomic)
    D, C, L, H, lh, desired, gridtype, ac, interp = case
    pls, S, off, table = _setup(D, C, L, H, lh, desired, 3, dtype, align_corners=ac)
    B = 4000
    x = _points(B, D, 4)
    rng = np.random.default_rng(9)
    grad = (rng.standard_normal((L, B, C)) * 0.1).astype(dtype)
    grad[:, 7] = 0                                     # zero-gradient rows are skipped
    _, dy = oracle.grid_encode_forward(x, table, off, D, C, L, S, H, True, gridtype, ac, interp, acc_mode=1)
    ge_ref, gi_ref = oracle.grid_encode_backward(grad, x, off, int(off[-1]), D, C, L, S, H, dy, gridtype, ac, interp)
    # the same (half-valued) gradients summed in fp32: what the binned path's fixed-point sums approximate (one rounding per row)
    ge_ref32 = oracle.grid_encode_backward(grad.astype(np.float32), x, off, int(off[-1]), D, C, L, S, H, None, gridtype, ac, interp)
    binned = atomic == "0" and D == 3 and C == 2 and gridtype == 0          # the shapes gb_check routes to the partition + LDS path
    tdt = torch.float32 if dtype == np.float32 else torch.float16
    xt, tt, ot = torch.from_numpy(x).cuda(), torch.from_numpy(table).cuda(), torch.from_numpy(off).cuda()
    be = _be()
    for bl in (False, True):
        g = torch.from_numpy(grad).cuda()
        if bl:
            g = g.permute(1, 0, 2).reshape(B, L * C).contiguous()
        ge = torch.zeros(int(off[-1]), C, dtype=tdt, device="cuda")
        gi = torch.zeros(B, D, dtype=tdt, device="cuda")
        be.grid_encode_backward(g, xt, tt, ot, ge, B, D, C, L, S, H, torch.from_numpy(dy.reshape(B, -1)).cuda(), gi, gridtype, ac, interp, grad_bl=bl)
        got, want = to_np(ge).astype(np.float32), ge_ref.astype(np.float32)
        if dtype == np.float32:
            np.testing.assert_allclose(got, want, atol=2e-5, rtol=1e-4)
            np.testing.assert_allclose(to_np(gi).astype(np.float32), gi_ref.astype(np.float32), atol=1e-3, rtol=1e-3)
        else:
            scale = np.abs(want).max()
            if binned:
                # exact fixed-point sums rounded to half once per row: against the fp32-summed oracle, like test_backward_ray_coherent
                want32 = ge_ref32.astype(np.float32)
                assert np.abs(got - want32).max() <= 2e-3 * np.abs(want32).max() + 1e-3, np.abs(got - want32).max()
            else:
                # half atomics round the running sum at every add (as the reference's __half2 atomicAdd does, which the oracle's fp16 mode
                # restates): order-dependent, so the bound is that of a handful of half roundings per row
                assert np.abs(got - want).max() <= 2e-2 * scale + 1e-3
            np.testing.assert_allclose(to_np(gi).astype(np.float32), gi_ref.astype(np.float32), atol=5e-2, rtol=5e-2)
        # checksum: interpolation weights sum to 1, so per level sum(grad_embeddings) == sum(grad) over in-range points
        inb = np.all((x >= 0) & (x <= 1), axis=1)
        if interp == 0:
            for l in range(L):
                want_sum = grad[l][inb].astype(np.float64).sum(0)
                got_sum = got[off[l]:off[l + 1]].astype(np.float64).sum(0)
                tol = 1e-3 if dtype == np.float32 else 0.5
                np.testing.assert_allclose(got_sum, want_sum, atol=tol)


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
@pytest.mark.parametrize("n_samples", [512, 100])
def test_backward_ray_coherent(dtype, n_samples):
    """Consecutive points = consecutive samples of a ray: neighbouring lanes share cells on the coarse levels, which is what the
    binned backward's run merging (csrc/gridencoder.hip, gb_run_flags) acts on. Ragged batch (not a multiple of 1024 or 16),
    out-of-range samples inside the runs, zero gradients, axis-parallel rays (very long runs)."""
    D, C, L, H, lh, desired, gridtype, ac, interp = CASES[0]
    pls, S, off, table = _setup(D, C, L, H, lh, desired, 3, dtype)
    rng = np.random.default_rng(21)
    n_rays = 9
    o = rng.random((n_rays, 1, 3)).astype(np.float32) * 0.2
    d = rng.random((n_rays, 1, 3)).astype(np.float32)
    d[1] = (1.0, 0.0, 0.0)                                    # axis-parallel: every coarse-level run hits the 16-lane cap
    d[2] = (0.0, 0.0, 0.0)                                    # a ray of identical points
    t = np.linspace(0.0, 0.85, n_samples, dtype=np.float32)[None, :, None]
    x = (o + d * t).reshape(-1, 3)[:-37].copy()               # ragged tail
    B = x.shape[0]
    x[5::97] = -0.25                                          # out-of-range samples break runs and emit nothing
    x[200:260, 1] = 1.5
    grad = (rng.standard_normal((L, B, C)) * 0.1).astype(dtype)
    grad[:, 300:340] = 0
    # expected sums in fp32 from the same (half-valued) gradients: the oracle's fp16 mode rounds its running sum at every add like the
    # reference's half2 atomics, which for 512 addends per row says more about that rounding than about the kernel under test
    ge_ref = oracle.grid_encode_backward(grad.astype(np.float32), x, off, int(off[-1]), D, C, L, S, H, None, gridtype, ac, interp)
    tdt = torch.float32 if dtype == np.float32 else torch.float16
    xt, tt, ot = torch.from_numpy(x).cuda(), torch.from_numpy(table).cuda(), torch.from_numpy(off).cuda()
    g = torch.from_numpy(grad).cuda().permute(1, 0, 2).reshape(B, L * C).contiguous()
    ge = torch.zeros(int(off[-1]), C, dtype=tdt, device="cuda")
    _be().grid_encode_backward(g, xt, tt, ot, ge, B, D, C, L, S, H, None, None, gridtype, ac, interp, grad_bl=True)
    got, want = to_np(ge).astype(np.float32), ge_ref.astype(np.float32)
    if dtype == np.float32:
        np.testing.assert_allclose(got, want, atol=2e-5, rtol=1e-4)
    else:
        scale = np.abs(want).max()
        assert np.abs(got - want).max() <= 2e-3 * scale + 1e-3
    inb = np.all((x >= 0) & (x <= 1), axis=1)
    for l in range(L):
        want_sum = grad[l][inb].astype(np.float64).sum(0)
        got_sum = got[off[l]:off[l + 1]].astype(np.float64).sum(0)
        np.testing.assert_allclose(got_sum, want_sum, atol=1e-3 if dtype == np.float32 else 0.5)
    # deterministic: a second launch gives the same bits (fixed-point / f64 LDS sums, ordered record ranges)
    ge2 = torch.zeros_like(ge)
    _be().grid_encode_backward(g, xt, tt, ot, ge2, B, D, C, L, S, H, None, None, gridtype, ac, interp, grad_bl=True)
    if dtype == np.float16:
        assert torch.equal(ge, ge2)


def test_grid_encode_autograd_and_module():
    """GridEncoder module: fp32 and autocast(fp16) paths, both kernel layouts, gradients w.r.t. table and inputs."""
    import os
    from focnerf_amd.gridencoder import GridEncoder
    torch.manual_seed(0)
    enc = GridEncoder(desired_resolution=2048).cuda()
    enc.embeddings.data.uniform_(-1, 1)
    x = (torch.rand(3000, 3, device="cuda") * 2 - 1).requires_grad_(True)
    outs = {}
    for lbc in ("0", "1"):
        os.environ["FOCNERF_GRID_LBC"] = lbc
        enc.zero_grad(); x.grad = None
        y = enc(x, bound=1)
        assert y.shape == (3000, 32) and y.dtype == torch.float32
        w = torch.linspace(-1, 1, 32, device="cuda")
        (y * w).sum().backward()
        outs[lbc] = (y.detach().clone(), enc.embeddings.grad.clone(), x.grad.clone())
    os.environ["FOCNERF_GRID_LBC"] = "0"
    assert torch.equal(outs["0"][0], outs["1"][0])
    assert torch.allclose(outs["0"][1], outs["1"][1], atol=1e-5)
    assert torch.allclose(outs["0"][2], outs["1"][2], atol=1e-3, rtol=1e-3)
    # oracle on the same data
    S = float(np.log2(enc.per_level_scale))
    xin = ((x.detach() + 1) / 2).cpu().numpy()
    ref = oracle.grid_encode_forward(xin, to_np(enc.embeddings), to_np(enc.offsets), 3, 2, 16, S, 16)
    assert np.array_equal(to_np(outs["0"][0]), np.transpose(ref, (1, 0, 2)).reshape(3000, 32))
    # finite-difference check of d/dx through dy_dx (linear interpolation is piecewise linear in x)
    with torch.no_grad():
        e = 1e-4
        xp = x.detach().clone(); xp[:, 0] += e
        xm = x.detach().clone(); xm[:, 0] -= e
        fd = ((enc(xp) - enc(xm)) * w).sum(-1) / (2 * e)
    good = (x.detach().abs() < 0.999).all(-1)
    rel = (fd[good] - outs["0"][2][good, 0]).abs() / (fd[good].abs() + 1.0)
    assert rel.median() < 2e-2
    # autocast: half table, half outputs, fp32 parameter gradient
    enc.zero_grad()
    with torch.autocast("cuda", dtype=torch.float16):
        yh = enc(x.detach(), bound=1)
    assert yh.dtype == torch.float16
    yh.float().sum().backward()
    assert enc.embeddings.grad.dtype == torch.float32
    refh = oracle.grid_encode_forward(xin, to_np(enc.embeddings).astype(np.float16), to_np(enc.offsets), 3, 2, 16, S, 16)
    assert_bits_equal(to_np(yh), np.ascontiguousarray(np.transpose(refh, (1, 0, 2)).reshape(3000, 32)), "autocast forward")
    # the module's backward hands the [B, L*C] gradient to the binned kernels as planes (one transpose) or as rows: the same records, the same sums
    g_planes = enc.embeddings.grad.clone()
    os.environ["FOCNERF_GRID_BWD_ROWS"] = "1"
    try:
        enc.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16):
            enc(x.detach(), bound=1).float().sum().backward()
    finally:
        del os.environ["FOCNERF_GRID_BWD_ROWS"]
    assert torch.equal(enc.embeddings.grad, g_planes) and g_planes.abs().sum() > 0


def test_grad_total_variation():
    from focnerf_amd.gridencoder import GridEncoder
    torch.manual_seed(1)
    enc = GridEncoder(num_levels=8, desired_resolution=256, log2_hashmap_size=15).cuda()
    enc.embeddings.data.uniform_(-1, 1)
    enc.embeddings.grad = torch.zeros_like(enc.embeddings)
    x = torch.rand(5000, 3, device="cuda") * 2 - 1
    with pytest.raises(ValueError):
        GridEncoder(num_levels=2, desired_resolution=32, log2_hashmap_size=10).cuda().grad_total_variation()
    enc.grad_total_variation(weight=1e-2, inputs=x, bound=1)
    S = float(np.log2(enc.per_level_scale))
    ref = oracle.grad_total_variation(((x + 1) / 2).cpu().numpy(), to_np(enc.embeddings), np.zeros_like(to_np(enc.embeddings)),
                                      to_np(enc.offsets), 1e-2, 3, 2, 8, S, 16)
    np.testing.assert_allclose(to_np(enc.embeddings.grad), ref, atol=1e-6, rtol=1e-4)


@pytest.mark.parametrize("D", [4, 5])
def test_module_and_total_variation_in_four_and_five_dimensions(D):
    """GridEncoder with input_dim 4 / 5 (the reference builds them for space-time grids): module forward + autograd and
    grad_total_variation against the oracle."""
    from focnerf_amd.gridencoder import GridEncoder
    torch.manual_seed(D)
    enc = GridEncoder(input_dim=D, num_levels=6, base_resolution=4, desired_resolution=32, log2_hashmap_size=13).cuda()
    enc.embeddings.data.uniform_(-1, 1)
    x = (torch.rand(3000, D, device="cuda") * 2 - 1).requires_grad_(True)
    y = enc(x, bound=1)
    assert y.shape == (3000, 12)
    S = float(np.log2(enc.per_level_scale))
    xn = to_np((x.detach() + 1) / 2)
    ref, ref_dy = oracle.grid_encode_forward(xn, to_np(enc.embeddings), to_np(enc.offsets), D, 2, 6, S, 4, True)
    assert np.array_equal(to_np(y), np.transpose(ref, (1, 0, 2)).reshape(3000, 12))
    gy = torch.randn(3000, 12, device="cuda") * 0.1
    y.backward(gy)
    g_lbc = np.ascontiguousarray(np.transpose(to_np(gy).reshape(3000, 6, 2), (1, 0, 2)))
    ge_ref, gi_ref = oracle.grid_encode_backward(g_lbc, xn, to_np(enc.offsets), enc.embeddings.shape[0], D, 2, 6, S, 4, ref_dy)
    np.testing.assert_allclose(to_np(enc.embeddings.grad), ge_ref, atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(to_np(x.grad), gi_ref / 2, atol=1e-3, rtol=1e-3)      # d/dx of (x + bound) / (2 bound)
    enc.embeddings.grad = torch.zeros_like(enc.embeddings)
    enc.grad_total_variation(weight=1e-2, inputs=x.detach(), bound=1)
    tv = oracle.grad_total_variation(xn, to_np(enc.embeddings), np.zeros_like(to_np(enc.embeddings)), to_np(enc.offsets), 1e-2, D, 2, 6, S, 4)
    np.testing.assert_allclose(to_np(enc.embeddings.grad), tv, atol=1e-6, rtol=1e-4)


def test_full_batch_properties():
    """BASELINE size: B = 4096 rays x 512 samples = 2 097 152 points, default 16-level table (fp16)."""
    from focnerf_amd.gridencoder import GridEncoder
    torch.manual_seed(2)
    enc = GridEncoder(desired_resolution=2048).cuda()
    enc.embeddings.data.uniform_(-1, 1)
    B = 4096 * 512
    x = torch.rand(B, 3, device="cuda") * 2 - 1
    with torch.autocast("cuda", dtype=torch.float16):
        y = enc(x)
        # a random subset against the oracle: rows are independent, so this is exact
        sel = torch.randperm(B, device="cuda")[:4096]
        S = float(np.log2(enc.per_level_scale))
        ref = oracle.grid_encode_forward(((x[sel] + 1) / 2).cpu().numpy(), to_np(enc.embeddings).astype(np.float16), to_np(enc.offsets), 3, 2, 16, S, 16)
        assert_bits_equal(to_np(y[sel]), np.ascontiguousarray(np.transpose(ref, (1, 0, 2)).reshape(4096, 32)), "full batch subset")
    # linearity in the table (fp32 path): T -> 2T doubles every output exactly (power-of-two scaling)
    y1 = enc(x)
    enc.embeddings.data.mul_(2)
    y2 = enc(x)
    enc.embeddings.data.mul_(0.5)
    assert torch.equal(y2, y1 * 2)
    del y1, y2
    # backward checksum at full size, fp32: per level, sum(grad_embeddings) == sum(grad)
    enc.zero_grad()
    yf = enc(x)
    g = torch.randn(B, 32, device="cuda") * 1e-3
    yf.backward(g)
    ge = enc.embeddings.grad
    off = to_np(enc.offsets)
    for l in range(16):
        got = ge[off[l]:off[l + 1]].double().sum(0)
        want = g[:, 2 * l:2 * l + 2].double().sum(0)
        assert torch.allclose(got, want, atol=2e-3), f"level {l}"


def test_beyond_baseline_size_and_render_view_size():
    """Sizes past the training batch: B = 6 M + 37 ray-ordered points in fp16 (1.6 G record slots: the 32-bit record indices of the
    binned backward are still in range, the ragged tail exercises the partial last workgroup), through the encoder -> MLP node."""
    from focnerf_amd.network import NeRFNetwork
    from focnerf_amd.field import hashgrid_mlp
    torch.manual_seed(5)
    m = NeRFNetwork(bound=1).cuda().train()
    m.encoder.embeddings.data.uniform_(-0.5, 0.5)
    n_rays, T = 11719, 512
    B = n_rays * T + 37                                              # 6 000 165
    gen = torch.Generator(device="cuda").manual_seed(1)
    o = torch.rand(n_rays + 1, 1, 3, device="cuda", generator=gen) * 0.4 - 0.2
    d = torch.nn.functional.normalize(torch.randn(n_rays + 1, 1, 3, device="cuda", generator=gen), dim=-1)
    t = torch.linspace(-1.0, 1.0, T, device="cuda")[None, :, None]
    x = (o + d * t).clamp(-1, 1).reshape(-1, 3)[:B].contiguous()
    with torch.autocast("cuda", dtype=torch.float16):
        h = hashgrid_mlp(m.encoder, m.sigma_net, x, 1)
        sel = torch.randint(0, B, (2048,), device="cuda", generator=gen)
        h_small = hashgrid_mlp(m.encoder, m.sigma_net, x[sel], 1)
    assert torch.equal(h[sel], h_small), "rows are independent of the batch they are evaluated in"
    g = (torch.randn(B, 16, device="cuda", generator=gen) * 1e-2).half()
    h.backward(g)
    ge = m.encoder.embeddings.grad
    assert torch.isfinite(ge).all() and ge.abs().max() > 0
    # same gradient from three slices of the batch (fixed-point sums per slice, added in fp32): linear in the batch
    ge_full = ge.clone()
    m.zero_grad(set_to_none=True)
    cuts = [0, 2_000_000, 4_100_000 + 11, B]
    for a, b in zip(cuts[:-1], cuts[1:]):
        with torch.autocast("cuda", dtype=torch.float16):
            hashgrid_mlp(m.encoder, m.sigma_net, x[a:b], 1).backward(g[a:b])
    scale = ge_full.abs().max().item()
    assert (m.encoder.embeddings.grad - ge_full).abs().max().item() <= 4e-3 * scale


def test_freq_encoder():
    from focnerf_amd.freqencoder import FreqEncoder
    torch.manual_seed(0)
    # (degree, rows, input dimension): ragged and whole 256-row tiles, one row, more tiles than workgroups of the grid-stride launch,
    # row widths C = D + 2 D deg below and above the 256-element stride of the tile walk (C = 27, 63, 39, 18, 25, 305)
    # (the reference takes any input dimension: 33, 100 and 3000 columns run with fewer rows per tile — 248, 81 and 2)
    for deg, B, D in [(4, 10007, 3), (10, 513, 3), (6, 1, 3), (4, 256 * 5, 2), (2, 70001, 5), (4, 256 * 2100 + 3, 3), (30, 300, 5), (2, 1000, 33), (1, 517, 100),
                      (1, 37, 3000)]:
        enc = FreqEncoder(D, deg)
        x = (torch.randn(B, D, device="cuda")).requires_grad_(True)
        y = enc(x)
        assert y.shape == (B, D + 2 * D * deg)
        ref = oracle.freq_encode_forward(to_np(x), deg)
        # sin(2^f x) at large arguments: ocml vs libm differ by a few ulp of the ARGUMENT reduction
        big = deg > 6
        np.testing.assert_allclose(to_np(y), ref, atol=(2e-5 if deg <= 10 else 1.0) if big else 2e-6)
        if deg > 10:                                   # 2^29 x: the argument itself has no fractional bits left; the identity columns and low octaves still match
            np.testing.assert_allclose(to_np(y)[:, : D + 2 * D * 6], ref[:, : D + 2 * D * 6], atol=2e-5)
        g = torch.randn_like(y)
        y.backward(g)
        gref = oracle.freq_encode_backward(to_np(g), to_np(y), D, deg)
        np.testing.assert_allclose(to_np(x.grad), gref, atol=1e-3 * 2.0 ** max(0, deg - 10), rtol=1e-4)


@pytest.mark.parametrize("atomic", ["0", "1"])
def test_backward_propagates_non_finite_gradients(atomic, monkeypatch):
    """An overflowed AMP step hands inf/NaN gradients to the encoder backward; torch's GradScaler skips the step only if it can SEE
    them in the parameter gradients. The reference's half2 atomics leave inf/NaN in the touched rows; so must the binned path
    (whose exact fixed-point sum has no encoding for them)."""
    monkeypatch.setenv("FOCNERF_GRID_ATOMIC", atomic)
    D, C, L, H, lh, desired, gridtype, ac, interp = CASES[0]
    pls, S, off, table = _setup(D, C, L, H, lh, desired, 3, np.float16)
    B = 3000
    x = _points(B, D, 4, oob=False)
    grad = (np.random.default_rng(1).standard_normal((B, L * C)) * 0.1).astype(np.float16)
    grad[100, 5] = np.inf          # level 2, channel 1
    grad[2000, 20] = np.nan        # level 10, channel 0
    xt, tt, ot = torch.from_numpy(x).cuda(), torch.from_numpy(table).cuda(), torch.from_numpy(off).cuda()
    ge = torch.zeros(int(off[-1]), C, dtype=torch.float16, device="cuda")
    _be().grid_encode_backward(torch.from_numpy(grad).cuda(), xt, tt, ot, ge, B, D, C, L, S, H, None, None, gridtype, ac, interp, grad_bl=True)
    got = to_np(ge).astype(np.float32)
    for l, n_bad_min in ((2, 1), (10, 1)):
        assert (~np.isfinite(got[off[l]:off[l + 1]])).sum() >= n_bad_min, f"level {l}: non-finite gradient was swallowed"
    for l in (0, 1, 3, 9, 11, 15):
        assert np.isfinite(got[off[l]:off[l + 1]]).all(), f"level {l} must stay finite"
    assert not torch.isfinite(ge).all()      # what GradScaler's unscale_ looks at


@pytest.mark.parametrize("layout", ["reference", "odd_offsets", "unaligned_table"])
def test_forward_row_pair_loads_and_their_fallbacks(layout):
    """fp16 C=2 D=3 forward without dy_dx: the two corners along x share one 8-byte load when their rows are neighbours (every even x
    on a hashed level, every even row on a dense one). Same bits as the oracle for (a) the reference's level offsets (multiples of 8:
    pairs on every level), (b) levels that start at odd rows / have odd sizes (no aligned pairs: per-level fallback, and a hashed level
    whose size is not a power of two), (c) a table whose base is only 4-byte aligned (whole launch falls back)."""
    D, C, L, H = 3, 2, 8, 16
    rng = np.random.default_rng(11)
    pls = 1.5
    S = float(np.log2(pls))
    res = [int(np.ceil(H * pls ** l - 1e-9)) for l in range(L)]
    if layout == "odd_offsets":
        sizes = [min((r + 1) ** 3, 100003) for r in res]          # dense levels with odd sizes, hashed levels of a prime size
    else:
        sizes = [int(np.ceil(min((r + 1) ** 3, 2 ** 16) / 8) * 8) for r in res]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    table = rng.uniform(-1, 1, (int(off[-1]) + 1, C)).astype(np.float16)
    B = 20000
    x = _points(B, D, 5)
    # scale as gridencoder.cu:112-113: exp2(level * S) * H - 1 -> resolution ceil(scale) + 1; the offsets above follow the same formula
    xt, ot = torch.from_numpy(x).cuda(), torch.from_numpy(off).cuda()
    tt_full = torch.from_numpy(table).cuda()
    if layout == "unaligned_table":
        tt, tab = tt_full[1:], table[1:]                            # base pointer 4 bytes past an aligned address
        assert tt.data_ptr() % 8 == 4
    else:
        tt, tab = tt_full[:-1], table[:-1]
    ref = oracle.grid_encode_forward(x, np.ascontiguousarray(tab), off, D, C, L, S, H, False, 0, False, 0, acc_mode=1)
    out = torch.empty(L, B, C, dtype=torch.float16, device="cuda")
    _be().grid_encode_forward(xt, tt, ot, out, B, D, C, L, S, H, None, 0, False, 0)
    assert_bits_equal(to_np(out), ref, f"outputs [L,B,C] ({layout})")


def test_precounted_backward_and_stale_tickets(monkeypatch):
    """The fused encoder->MLP node runs the backward's count pass during the forward, on a side stream; a ticket ties the workspace
    header to that forward. Gradients must equal the plain path's, also when another forward or another backward used the workspace in
    between (stale ticket -> the backward counts again)."""
    from focnerf_amd.field import hashgrid_mlp
    from focnerf_amd.network import NeRFNetwork
    torch.manual_seed(3)
    m = NeRFNetwork(bound=1).cuda().train()
    m.encoder.embeddings.data.uniform_(-0.5, 0.5)
    xa = torch.rand(5000, 3, device="cuda") * 2 - 1
    xb = torch.rand(7000, 3, device="cuda") * 2 - 1

    def grads(order, pre):
        monkeypatch.setenv("FOC_GRID_PRECOUNT", pre)
        m.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16):
            ha = hashgrid_mlp(m.encoder, m.sigma_net, xa, 1)
            hb = hashgrid_mlp(m.encoder, m.sigma_net, xb, 1)        # second forward: overwrites the header counted for `ha`
        la, lb = (ha.float() ** 2).sum(), (hb.float().abs()).sum()
        first, second = (la, lb) if order == "ab" else (lb, la)
        first.backward()
        second.backward()
        torch.cuda.synchronize()
        return m.encoder.embeddings.grad.clone(), m.sigma_net.weights.grad.clone()

    ref = grads("ab", "0")
    for order in ("ab", "ba"):
        got = grads(order, "1")
        for a, b in zip(ref, got):
            scale = a.abs().max().item()
            assert scale > 0 and torch.isfinite(b).all()
            assert (a.float() - b.float()).abs().max().item() <= 2e-3 * scale


def test_no_grad_grid_encode_leaves_the_backward_state_alone():
    """The public op under torch.no_grad() (evaluation through the drop-in path): `needs_input_grad` stays True for an nn.Parameter table there,
    but the call must not run the backward's count pass — it neither takes a precount ticket (which would invalidate a pending training
    forward's) nor touches the grid_bwd scratch; and a training forward's ticket survives an evaluation call made before its backward."""
    from focnerf_amd import backend
    from focnerf_amd.gridencoder import GridEncoder
    torch.manual_seed(5)
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048).cuda().train()
    enc.embeddings.data.uniform_(-0.5, 0.5)
    xa = torch.rand(6000, 3, device="cuda") * 2 - 1
    xe = torch.rand(9000, 3, device="cuda") * 2 - 1
    _, st = backend._gridencoder._pre_state(xa.device)
    calls = []
    real = backend._gridencoder.grid_encode_forward_counted
    backend._gridencoder.grid_encode_forward_counted = staticmethod(lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    try:
        with torch.autocast("cuda", dtype=torch.float16):
            ya = enc(xa, 1)                                   # training forward: counted, takes a ticket
            before = dict(st)
            scratch_before = {k: v[0].data_ptr() for k, v in backend._scratch._bufs.items()}
            assert len(calls) == 1 and before["key"] is not None
            with torch.no_grad():
                ye = enc(xe, 1)                               # evaluation in between
            assert len(calls) == 1, "a no_grad grid_encode ran the backward's count pass"
            assert dict(st) == before, "a no_grad grid_encode changed the precount ticket"
            assert {k: v[0].data_ptr() for k, v in backend._scratch._bufs.items()} == scratch_before
        (ya.float() ** 2).sum().backward()
        g_counted = enc.embeddings.grad.clone()
    finally:
        backend._gridencoder.grid_encode_forward_counted = staticmethod(real)
    assert not ye.requires_grad and torch.isfinite(ye).all()
    # the same gradient as a step without the evaluation call in between
    enc.embeddings.grad = None
    with torch.autocast("cuda", dtype=torch.float16):
        yb = enc(xa, 1)
    (yb.float() ** 2).sum().backward()
    assert torch.equal(ya, yb) and torch.equal(g_counted, enc.embeddings.grad)


@pytest.mark.parametrize("standalone", [False, True])
@pytest.mark.parametrize("B,L", [(1, 16), (1000, 16), (70001, 16), (5000, 2), (3000, 5)])
def test_counted_forward_and_counted_backward_match_the_plain_calls(B, L, standalone):
    """foc_grid_encode_forward_counted (count pass riding in the forward launch) / foc_grid_encode_backward_count (its own launch) followed
    by foc_grid_encode_backward_binned_counted: same encoding bits and same gradient bits as the plain forward + binned backward."""
    D, C, H = 3, 2, 16
    pls, S, off, table = _setup(D, C, L, H, 19, 2048 if L == 16 else 256, 1, np.float16)
    rng = np.random.default_rng(B)
    x = _points(B, D, 9) if B > 8 else rng.random((B, D)).astype(np.float32)
    g = (rng.standard_normal((L, B, C)) * 0.1).astype(np.float16)
    xt, tt, ot, gt = (torch.from_numpy(a).cuda() for a in (x, table, off, g))
    be = _be()
    out0 = torch.empty(L, B, C, dtype=torch.float16, device="cuda")
    be.grid_encode_forward(xt, tt, ot, out0, B, D, C, L, S, H, None, 0, False, 0)
    ge0 = torch.zeros_like(tt)
    be.grid_encode_backward(gt, xt, tt, ot, ge0, B, D, C, L, S, H, None, None, 0, False, 0)
    out1 = torch.empty_like(out0)
    ticket = be.grid_encode_forward_counted(xt, tt, ot, out1, B, D, C, L, S, H, 0, False, 0, standalone=standalone)
    assert ticket is not None
    assert torch.equal(out0, out1)
    ge1 = torch.zeros_like(tt)
    be.grid_encode_backward(gt, xt, tt, ot, ge1, B, D, C, L, S, H, None, None, 0, False, 0, precount=ticket)
    # segments with more than 65 536 records (the coarse levels at the largest B) are summed in several chunks whose fp16 partial sums meet
    # in half2 atomics: their order, hence the last bit, is not fixed from run to run
    same = (lambda a, b: torch.equal(a, b)) if B <= 1000 else (lambda a, b: assert_half_close(to_np(b), to_np(a), ulps=2.0, atol=4 * 2.0 ** -10 * float(a.abs().max()), what="grad") is None)
    assert same(ge0, ge1)
    # a spent ticket is not honoured twice (the backward counts again) and still gives the same result
    ge2 = torch.zeros_like(tt)
    be.grid_encode_backward(gt, xt, tt, ot, ge2, B, D, C, L, S, H, None, None, 0, False, 0, precount=ticket)
    assert same(ge0, ge2)


def test_tables_beyond_the_binned_limit_take_the_atomic_kernel():
    """log2_hashmap_size = 20: a level has 2^20 rows, more than the 64 x 8192 the binned backward partitions; the wrapper must route such
    tables to the scattered-atomic kernel (and the counted forward must decline) instead of failing."""
    from focnerf_amd.gridencoder import GridEncoder
    from focnerf_amd.field import hashgrid_mlp
    from focnerf_amd.ffmlp import FFMLP
    torch.manual_seed(0)
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=20, desired_resolution=2048).cuda()
    enc.embeddings.data.uniform_(-0.5, 0.5)
    mlp = FFMLP(32, 16, 64, 2).cuda().train()
    x = torch.rand(3000, 3, device="cuda") * 2 - 1
    with torch.autocast("cuda", dtype=torch.float16):
        h = hashgrid_mlp(enc, mlp, x, 1)
    (h.float() ** 2).sum().backward()
    g = enc.embeddings.grad
    assert g is not None and torch.isfinite(g).all() and g.abs().max() > 0
    # against the public two-node path (grid_encode -> FFMLP)
    g_fused = g.clone(); enc.embeddings.grad = None; mlp.weights.grad = None
    with torch.autocast("cuda", dtype=torch.float16):
        h2 = mlp.forward_padded(enc(x, 1))
    (h2.float() ** 2).sum().backward()
    assert torch.equal(h, h2)
    scale = g_fused.abs().max().item()
    assert (g_fused - enc.embeddings.grad).abs().max().item() <= 2e-2 * scale


@pytest.mark.parametrize("what", ["tiled", "very_fine"])
def test_grids_outside_the_two_corner_record_bound_take_the_atomic_kernel(what):
    """The binned backward's staging holds 5 two-corner records per point and level, a bound that holds for hash grids with resolutions
    below 8191 (csrc/gridencoder.hip, gb_check). Tiled D=3/C=2 grids and finer hash grids must be routed to the atomic kernel by the
    wrapper and give the oracle's gradient."""
    D, C, L, H = 3, 2, (4 if what == "tiled" else 12), 16
    gridtype = 1 if what == "tiled" else 0
    desired = 512 if what == "tiled" else 16384
    pls, S, off, table = _setup(D, C, L, H, 15, desired, 1, np.float32)
    B = 3000
    x = _points(B, D, 8, oob=False)
    grad = (np.random.default_rng(2).standard_normal((L, B, C)) * 0.1).astype(np.float32)
    ref = oracle.grid_encode_backward(grad, x, off, int(off[-1]), D, C, L, S, H, None, gridtype, False, 0)
    xt, tt, ot, gt = (torch.from_numpy(a).cuda() for a in (x, table, off, grad))
    ge = torch.zeros_like(tt)
    _be().grid_encode_backward(gt, xt, tt, ot, ge, B, D, C, L, S, H, None, None, gridtype, False, 0)
    got = to_np(ge)
    scale = np.abs(ref).max()
    assert scale > 0 and np.abs(got - ref).max() <= 1e-5 * scale
    out = torch.empty(L, B, C, dtype=torch.float32, device="cuda")
    assert _be().grid_encode_forward_counted(xt, tt, ot, out, B, D, C, L, S, H, gridtype, False, 0) is None


def test_encoder_mlp_node_replays_in_a_hip_graph_with_new_data():
    """Forward (count pass riding along) + backward of the encoder -> MLP node captured once and replayed on other inputs: the record
    counts, the table gradient and the weight gradient must follow the data of each replay. Guards the zero fills inside the library:
    as hipMemsetAsync nodes they came back with a garbage fill value from the second replay on (ROCm 7.2), the header held ~5 * 10^11
    'records' and the reduce read out of bounds; they are kernels now (csrc/common.h, foc_zero_async)."""
    from focnerf_amd.field import hashgrid_mlp
    from focnerf_amd.network import NeRFNetwork
    torch.manual_seed(1)
    m = NeRFNetwork(bound=1).cuda().train()
    m.encoder.embeddings.data.uniform_(-0.5, 0.5)
    B = 6000
    xs = [torch.rand(B, 3, device="cuda") * 2 - 1 for _ in range(4)]

    def step(x):
        m.encoder.embeddings.grad = None
        m.sigma_net.weights.grad = None
        with torch.autocast("cuda", dtype=torch.float16):
            h = hashgrid_mlp(m.encoder, m.sigma_net, x, 1)
        (h.float() ** 2).sum().backward()
        return m.encoder.embeddings.grad, m.sigma_net.weights.grad

    want = []
    for x in xs:
        ge, gw = step(x)
        want.append((ge.clone(), gw.clone()))
    static_x = xs[0].clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step(static_x)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ge_s, gw_s = step(static_x)
    for rounds in range(2):                                  # every input twice: nothing may carry over between replays
        for x, (ge, gw) in zip(xs, want):
            static_x.copy_(x)
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(ge_s, ge), f"table gradient, round {rounds}"
            scale = gw.abs().max().item()
            assert (gw_s.float() - gw.float()).abs().max().item() <= 2e-3 * scale, f"weight gradient, round {rounds}"


def test_factored_records_agree_with_the_two_corner_records(lib_option):
    """Unmerged hashed levels send a pair of corners as 8 bytes {row, jb, fx, half2 p} and the reduce rebuilds (1 - fx) p and fx p
    (csrc/gridencoder.hip, k_gbin_scatter_pms / k_gbin_reduce). Against the 12-byte form (both addends rounded to half by the scatter):
    p is rounded once before the split and fx carries 15 bits, so an addend moves by at most ~1 half-ulp; both forms are deterministic,
    keep the per-level checksum, and agree on which rows receive anything."""
    D, C, L, H, lh, desired, gridtype, ac, interp = CASES[0]
    pls, S, off, table = _setup(D, C, L, H, lh, desired, 3, np.float16)
    B = 6000                                          # 4 B records per level at most: every (level, segment) fits ONE reduce chunk of 32768 records, so
    x = _points(B, D, 11)                             # each row is rounded to half once (several chunks add their half-rounded partial sums in arrival order)
    rng = np.random.default_rng(5)
    grad = (rng.standard_normal((L, B, C)) * 0.25).astype(np.float16)
    xt, tt, ot = torch.from_numpy(x).cuda(), torch.from_numpy(table).cuda(), torch.from_numpy(off).cuda()
    g = torch.from_numpy(grad).cuda()
    res = {}
    for mode in ("0", "1", "1"):
        lib_option("FOC_GB_FACTORED", int(mode))
        ge = torch.zeros(int(off[-1]), C, dtype=torch.float16, device="cuda")
        _be().grid_encode_backward(g, xt, tt, ot, ge, B, D, C, L, S, H, None, None, gridtype, ac, interp, grad_bl=False)
        if mode in res:
            assert torch.equal(res[mode], ge), "the factored form is deterministic"
        res[mode] = ge
    a, b = to_np(res["0"]).astype(np.float32), to_np(res["1"]).astype(np.float32)
    assert not np.array_equal(a, b), "the switch must reach the kernels (the finest levels of this grid are factored)"
    assert np.array_equal(a != 0, b != 0) or np.abs(a - b)[(a != 0) != (b != 0)].max() < 2.0 ** -14
    # a row of a fine level receives a handful of addends of magnitude <= 0.25 * |w|: a few half-ulps of the row's value
    tol = 4 * 2.0 ** -11 * np.maximum(np.abs(a), np.abs(b)) + 4 * 2.0 ** -14
    assert (np.abs(a - b) <= tol).all(), float((np.abs(a - b) - tol).max())
    inb = np.all((x >= 0) & (x <= 1), axis=1)
    for l in range(L):
        want = grad[l][inb].astype(np.float64).sum(0)
        np.testing.assert_allclose(b[off[l]:off[l + 1]].astype(np.float64).sum(0), want, atol=0.5)


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
def test_levels_of_2_to_the_23_rows_take_the_generic_index_path(dtype):
    """log2_hashmap_size = 23, two levels (resolutions 256 and 512: both hashed, 2^23 rows each): beyond the 2^22 rows of the forward's
    24-bit index arithmetic (csrc/gridencoder.hip ge_forward_hash3) — the 32-bit byte-offset form with full multiplies for fp16 tables,
    ge_forward_one for fp32 — and beyond the 64 x 8192 rows the binned backward partitions (scattered-atomic kernel). Forward bit-exact
    against the oracle in both output layouts, backward within the atomic path's tolerances, per-level checksums."""
    from focnerf_amd._lib import lib
    D, C, L, H, lh, desired = 3, 2, 2, 256, 23, 512
    pls, S, off, table = _setup(D, C, L, H, lh, desired, 11, dtype)
    assert [int(off[i + 1] - off[i]) for i in range(L)] == [1 << 23, 1 << 23]
    assert lib.foc_grid_forward_index_path(1 << 23, 256, 0) == 1 and lib.foc_grid_forward_index_path(1 << 23, 512, 1 << 23) == 1
    B = 6000
    x = _points(B, D, 12)
    ref = oracle.grid_encode_forward(x, table, off, D, C, L, S, H, False, 0, False, 0, acc_mode=1)
    tdt = torch.float32 if dtype == np.float32 else torch.float16
    xt, tt, ot = torch.from_numpy(x).cuda(), torch.from_numpy(table).cuda(), torch.from_numpy(off).cuda()
    be = _be()
    out = torch.empty(L, B, C, dtype=tdt, device="cuda")
    be.grid_encode_forward(xt, tt, ot, out, B, D, C, L, S, H, None, 0, False, 0)
    assert_bits_equal(to_np(out), ref, "outputs [L,B,C]")
    out2 = torch.empty(B, L * C, dtype=tdt, device="cuda")
    be.grid_encode_forward(xt, tt, ot, out2, B, D, C, L, S, H, None, 0, False, 0, out_bl=True)
    assert np.array_equal(to_np(out2), np.transpose(ref, (1, 0, 2)).reshape(B, L * C))
    assert np.abs(ref.astype(np.float32)).max() > 0.5
    # the counted forward declines (no binned backward for these tables) and the backward takes the atomic kernel
    assert be.grid_encode_forward_counted(xt, tt, ot, out, B, D, C, L, S, H, 0, False, 0) is None
    grad = (np.random.default_rng(13).standard_normal((L, B, C)) * 0.1).astype(dtype)
    ge_ref = oracle.grid_encode_backward(grad, x, off, int(off[-1]), D, C, L, S, H, None, 0, False, 0)
    ge = torch.zeros(int(off[-1]), C, dtype=tdt, device="cuda")
    be.grid_encode_backward(torch.from_numpy(grad).cuda(), xt, tt, ot, ge, B, D, C, L, S, H, None, None, 0, False, 0, grad_bl=False)
    got, want = to_np(ge).astype(np.float32), ge_ref.astype(np.float32)
    if dtype == np.float32:
        np.testing.assert_allclose(got, want, atol=2e-5, rtol=1e-4)
    else:
        assert np.abs(got - want).max() <= 2e-2 * np.abs(want).max() + 1e-3
    inb = np.all((x >= 0) & (x <= 1), axis=1)
    for l in range(L):
        np.testing.assert_allclose(got[off[l]:off[l + 1]].astype(np.float64).sum(0), grad[l][inb].astype(np.float64).sum(0), atol=1e-3 if dtype == np.float32 else 0.5)


@pytest.mark.parametrize("dtype", [np.float16, np.float32])
def test_scatter_tail_split_is_bitwise_the_whole_tile_form(dtype, lib_option):
    """The last partial round of scatter workgroups is dealt out by level group (csrc/gridencoder.hip k_gbin_scatter_pms: FOC_GB_TAIL_SPLIT
    workgroups per tile, 16 / 8 / 4 / 2 levels each): the same record ranges, the same records, the same sums as whole tiles
    (FOC_GB_TAIL_SPLIT=1) — bit for bit, for every split the launch rule can pick, ragged last tile and clustered points included."""
    D, C, L, H, lh, desired, gridtype, ac, interp = CASES[0]
    pls, S, off, table = _setup(D, C, L, H, lh, desired, 3, dtype)
    B = 5 * 1024 + 77                                  # six tiles, the last one ragged; every one of them is "tail" on a 256-CU chip
    x = _points(B, D, 31)
    x[::5] = x[::5] * 0.03 + 0.45                       # clustered points: uneven segments
    grad = (np.random.default_rng(32).standard_normal((L, B, C)) * 0.25).astype(dtype)
    xt, tt, ot, g = torch.from_numpy(x).cuda(), torch.from_numpy(table).cuda(), torch.from_numpy(off).cuda(), torch.from_numpy(grad).cuda()
    tdt = torch.float16 if dtype == np.float16 else torch.float32
    res = {}
    for split in (1, 2, 4, 8, 16):
        lib_option("FOC_GB_TAIL_SPLIT", split)
        ge = torch.zeros(int(off[-1]), C, dtype=tdt, device="cuda")
        _be().grid_encode_backward(g, xt, tt, ot, ge, B, D, C, L, S, H, None, None, gridtype, ac, interp, grad_bl=False)
        res[split] = ge
    for split in (2, 4, 8, 16):
        assert torch.equal(res[split], res[1]), split
    assert res[1].abs().max() > 0
