"""CPU: pin the oracle — against the golden vectors produced by the imported reference
(tests/golden/*.npz, see make_golden.py) and against known answers derivable from the reference's
source text. Everything the reference itself cannot run on CPU stays
"parity unpinned" beyond these checks (oracle/oracle.c header)."""
import math
import os

import numpy as np
import pytest

import oracle


# ---------------------------------------------------------------- golden: reference NeRFRenderer.run
@pytest.mark.parametrize("name", ["run_foc.npz", "run_foc_b2.npz"])
def test_fixed_step_composite_matches_reference_run(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name))
    # R1 restatement reproduces the near/far the reference run() consumed (trivially: the stub was the oracle),
    # and is checked independently against the closed-form slab test below.
    nears, fars = oracle.near_far_from_aabb(g["rays_o"], g["rays_d"], g["aabb"], float(g["min_near"]))
    assert np.array_equal(nears, g["nears"]) and np.array_equal(fars, g["fars"])
    img4, depth, w = oracle.composite_fixed_steps(g["sigmas"], g["rgbs"], nears, fars, bg=1.0, clamp01=False, want_weights=True)
    hit = nears < 1e30
    assert hit.sum() > 0.7 * hit.size and (~hit).sum() > 0     # fixture holds rays that miss the box
    np.testing.assert_allclose(img4[:, :3], g["image"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(w.sum(-1), g["weights_sum"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(depth[hit], g["depth"][hit], rtol=0, atol=2e-6)
    # rays that miss the box: depth is 0*NaN = NaN in the reference, and in the oracle
    assert np.isnan(g["depth"][~hit]).all() and np.isnan(depth[~hit]).all()
    np.testing.assert_allclose(img4[~hit, :3], 1.0)             # pure background


def test_trunc_exp_matches_reference(golden_dir):
    import torch
    from focnerf_amd.activation import trunc_exp
    g = np.load(os.path.join(golden_dir, "trunc_exp.npz"))
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = trunc_exp(x)
    y.backward(torch.from_numpy(g["gy"]))
    np.testing.assert_array_equal(y.detach().numpy(), g["y"])
    np.testing.assert_array_equal(x.grad.numpy(), g["gx"])


# ---------------------------------------------------------------- known answers from the source text
def test_morton_known_answers_and_round_trip():
    # __expand_bits spreads bit i to bit 3i (raymarching.cu:56-63)
    c = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [2, 0, 0], [127, 127, 127], [5, 3, 6], [1023, 0, 0]], np.int32)
    m = oracle.morton3D(c)
    assert m[0] == 1 and m[1] == 2 and m[2] == 4 and m[3] == 8
    assert m[4] == 2 ** 21 - 1
    assert m[6] == 0x09249249                       # ten bits of x on every third bit
    # (5,3,6): x=101b y=011b z=110b -> bits (z y x) per level: l0: z0=0 y0=1 x0=1 -> 011b; l1: 1 1 0 -> 110b; l2: 1 0 1 -> 101b
    assert m[5] == (0b101 << 6) | (0b110 << 3) | 0b011
    rng = np.random.default_rng(0)
    c = rng.integers(0, 128, (4096, 3), dtype=np.int32)
    assert np.array_equal(oracle.morton3D_invert(oracle.morton3D(c)), c)
    idx = rng.integers(0, 128 ** 3, 4096, dtype=np.int32)
    assert np.array_equal(oracle.morton3D(oracle.morton3D_invert(idx)), idx)


def test_packbits_matches_numpy():
    rng = np.random.default_rng(1)
    grid = rng.random(8 * 1000).astype(np.float32)
    got = oracle.packbits(grid, 0.5)
    want = np.packbits((grid > 0.5).reshape(-1, 8), axis=1, bitorder="little").reshape(-1)
    assert np.array_equal(got, want)


def test_near_far_slab_closed_form():
    # axis-aligned ray through the unit box from outside: near = 1, far = 3 along +x from x=-2
    o = np.array([[-2, 0.1, 0.2], [0, 0, 0], [-2, 5, 0]], np.float32)
    d = np.array([[1, 0, 0], [0, 0, 1], [1, 0, 0]], np.float32)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    n, f = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    assert n[0] == 1.0 and f[0] == 3.0
    assert n[1] == np.float32(0.2) and f[1] == 1.0            # origin inside the box: min_near clamp
    fmax = np.finfo(np.float32).max
    assert n[2] == fmax and f[2] == fmax                       # miss


def test_level_offsets_table():
    """Rows per level for the default hash grid (SURVEY.md §8 header, from grid.py:117-131)."""
    from focnerf_amd.gridencoder import level_offsets
    for bound, rows in [(1, [4920, 13824, 32768, 85184, 216000] + [524288] * 11),
                        (2, [4920, 15632, 42880, 125000, 373248] + [524288] * 11)]:
        pls = np.exp2(np.log2(2048 * bound / 16) / 15)
        off = level_offsets(3, 16, pls, 16, 19)
        assert list(np.diff(off)) == rows
    assert off[-1] == 6328848


def test_grid_index_dense_vs_hash_and_primes():
    # one point per level; compare the oracle's corner rows against an independent Python evaluation
    pls = np.exp2(np.log2(2048 / 16) / 15)
    S = float(np.log2(pls))
    from focnerf_amd.gridencoder import level_offsets
    off = level_offsets(3, 16, pls, 16, 19)
    table = np.zeros((off[-1], 2), np.float32)
    x = np.array([[0.3, 0.6, 0.9]], np.float32)
    primes = [1, 2654435761, 805459861]
    for level in [0, 4, 5, 15]:
        scale, res = oracle.grid_level_params(level, S, 16)
        assert res == int(math.ceil(scale)) + 1
        size = int(off[level + 1] - off[level])
        pos = x[0].astype(np.float64) * scale + 0.5
        pg = np.floor(np.float32(x[0] * np.float32(scale) + np.float32(0.5))).astype(np.int64)
        stride, dense, s_exceeds = 1, 0, False
        for dd in range(3):
            if stride <= size:
                dense += int(pg[dd]) * stride
                stride *= res + 1
        use_hash = stride > size
        h = 0
        for dd in range(3):
            h ^= (int(pg[dd]) * primes[dd]) & 0xFFFFFFFF
        row = (h if use_hash else dense & 0xFFFFFFFF) % size
        assert use_hash == (level >= 5)                      # levels 0-4 dense, 5-15 hashed
        table[:] = 0
        table[off[level] + row, 0] = 1.0                     # mark corner (0,0,0)
        out = oracle.grid_encode_forward(x, table, off, 3, 2, 16, S, 16)
        frac = np.float32(x[0] * np.float32(scale) + np.float32(0.5)) - pg.astype(np.float32)
        w000 = np.prod(1 - frac)
        assert abs(out[level, 0, 0] - w000) < 1e-6 and out[level, 0, 1] == 0
        assert np.count_nonzero(out) == 1


def test_march_constants_and_simple_march():
    # dt_min = 2*sqrt(3)/max_steps, dt_max = 2*sqrt(3)*2^(C-1)/H (raymarching.cu:345-346)
    H, C, max_steps = 128, 1, 1024
    grid = np.full(C * H ** 3 // 8, 0xFF, np.uint8)            # everything occupied
    o = np.array([[-2.0, 0.0, 0.0]], np.float32)
    d = np.array([[1.0, 0.0, 0.0]], np.float32)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    n, f = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    xyzs, dirs, deltas, rays, counter = oracle.march_rays_train(o, d, grid, 1.0, 0.0, max_steps, C, H, 1024, n, f, np.zeros(1, np.float32))
    dt_min = np.float32(2 * np.float32(1.7320508075688772) / np.float32(max_steps))
    cnt = int(rays[0, 2])
    assert counter[0] == cnt and counter[1] == 1 and rays[0, 0] == 0 and rays[0, 1] == 0
    assert abs(cnt - 2.0 / dt_min) <= 1                       # dt_gamma = 0: constant step dt_min through the 2-unit box
    assert np.all(deltas[:cnt, 0] == dt_min)
    assert np.all(xyzs[cnt:] == 0)
    # empty grid: no samples at all
    xyzs, dirs, deltas, rays, counter = oracle.march_rays_train(o, d, np.zeros_like(grid), 1.0, 0.0, max_steps, C, H, 1024, n, f,
                                                                np.zeros(1, np.float32))
    assert counter[0] == 0 and rays[0, 2] == 0


def test_composite_constant_sigma_closed_form():
    # constant sigma: weights_sum = 1 - exp(-sigma * sum(dt))
    M = 50
    sig = np.full(M, 0.7, np.float32)
    rgb = np.full((M, 3), 0.25, np.float32)
    deltas = np.full((M, 2), 0.01, np.float32)
    rays = np.array([[0, 0, M]], np.int32)
    ws, depth, image = oracle.composite_rays_train_forward(sig, rgb, deltas, rays, 1, 1e-4)
    want = 1 - math.exp(-0.7 * 0.01 * M)
    assert abs(ws[0] - want) < 1e-6
    np.testing.assert_allclose(image[0], 0.25 * want, atol=1e-6)
    # early termination: opaque first sample ends the ray (T < T_thresh after it)
    sig[0] = 1e4
    deltas[0, 0] = 1.0
    ws, depth, image = oracle.composite_rays_train_forward(sig, rgb, deltas, rays, 1, 1e-4)
    assert ws[0] == 1.0


def test_composite_backward_matches_finite_differences():
    rng = np.random.default_rng(3)
    M = 24
    sig = rng.random(M).astype(np.float32) * 3
    rgb = rng.random((M, 3)).astype(np.float32)
    deltas = np.stack([rng.random(M) * 0.05 + 0.01, rng.random(M) * 0.05 + 0.01], -1).astype(np.float32)
    rays = np.array([[0, 0, M]], np.int32)
    gws = np.array([0.3], np.float32)
    gim = np.array([[0.5, -0.2, 0.9]], np.float32)
    ws, depth, image = oracle.composite_rays_train_forward(sig, rgb, deltas, rays, 1, 0.0)
    gs, gc = oracle.composite_rays_train_backward(gws, gim, sig, rgb, deltas, rays, ws, image, 0.0)

    def loss(s, c):
        w, _, im = oracle.composite_rays_train_forward(s, c, deltas, rays, 1, 0.0)
        return float(gws[0] * np.float64(w[0]) + (gim[0].astype(np.float64) * im[0].astype(np.float64)).sum())
    for i in [0, 5, 23]:
        e = 1e-2
        sp, sm = sig.copy(), sig.copy()
        sp[i] += e; sm[i] -= e
        fd = (loss(sp, rgb) - loss(sm, rgb)) / (2 * e)
        assert abs(fd - gs[i]) < 2e-3
        cp, cm = rgb.copy(), rgb.copy()
        cp[i, 1] += e; cm[i, 1] -= e
        fd = (loss(sig, cp) - loss(sig, cm)) / (2 * e)
        assert abs(fd - gc[i, 1]) < 2e-3


def test_freq_output_order():
    x = np.array([[0.1, -0.2, 0.3]], np.float32)
    out = oracle.freq_encode_forward(x, 2)                     # [x, sin(x), cos(x), sin(2x), cos(2x)]
    want = np.concatenate([x[0], np.sin(x[0]), np.cos(x[0]), np.sin(2 * x[0]), np.cos(2 * x[0])])
    np.testing.assert_allclose(out[0], want, atol=1e-6)
    g = np.random.default_rng(0).standard_normal((1, 15)).astype(np.float32)
    gi = oracle.freq_encode_backward(g, out, 3, 2)
    want_g = g[0, :3] + g[0, 3:6] * np.cos(x[0]) - g[0, 6:9] * np.sin(x[0]) + 2 * (g[0, 9:12] * np.cos(2 * x[0]) - g[0, 12:15] * np.sin(2 * x[0]))
    np.testing.assert_allclose(gi[0], want_g, atol=1e-5)


def test_half_conversion_matches_numpy():
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.standard_normal(20000) * 10 ** rng.uniform(-9, 5, 20000), [0.0, -0.0, 65504, 65519.9, 65520, 1e-8, 5.96e-8, 2.98e-8, 2.9802322e-8,
                        np.inf, -np.inf]]).astype(np.float32)
    with np.errstate(over="ignore"):
        want = x.astype(np.float16)
    got = oracle.f2h(x)
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))
    allh = np.arange(65536, dtype=np.uint16).view(np.float16)
    back = oracle.h2f(allh)
    ref = allh.astype(np.float32)
    ok = np.isnan(ref) | (back == ref)
    assert ok.all()


def test_ffmlp_oracle_matches_float_matmul():
    rng = np.random.default_rng(7)
    B, I, Hd, nl = 128, 32, 64, 2
    W = (rng.uniform(-1, 1, Hd * (I + Hd * (nl - 1) + 16)) * math.sqrt(3 / Hd)).astype(np.float16)
    x = rng.standard_normal((B, I)).astype(np.float16)
    out, fb = oracle.ffmlp_forward(x, W, I, Hd, nl)
    W0 = W[:Hd * I].reshape(Hd, I).astype(np.float32)
    W1 = W[Hd * I:Hd * I + Hd * Hd].reshape(Hd, Hd).astype(np.float32)
    Wo = W[Hd * I + Hd * Hd:].reshape(16, Hd).astype(np.float32)
    a0 = np.maximum(x.astype(np.float32) @ W0.T, 0).astype(np.float16)
    a1 = np.maximum(a0.astype(np.float32) @ W1.T, 0).astype(np.float16)
    y = (a1.astype(np.float32) @ Wo.T).astype(np.float16)
    # identical up to fp32 summation order before the single fp16 rounding
    assert np.abs(fb[0].astype(np.float32) - a0.astype(np.float32)).max() <= 2e-3 * np.abs(a0.astype(np.float32)).max()
    assert np.abs(out.astype(np.float32) - y.astype(np.float32)).max() <= 4e-3 * max(1.0, np.abs(y.astype(np.float32)).max())
    # the WMMA-like half-accumulate model stays within fp16 accumulation error of it
    out_h = oracle.ffmlp_forward(x, W, I, Hd, nl, training=False, acc_mode=1)
    assert np.abs(out_h.astype(np.float32) - out.astype(np.float32)).max() < 5e-2
    # backward: dW and dX against float matmuls
    g = rng.standard_normal((B, 16)).astype(np.float16) * np.float16(0.1)
    gw, gi, bb = oracle.ffmlp_backward(g, x, W, fb, I, Hd, nl)
    d1 = ((g.astype(np.float32) @ Wo) * (a1 > 0)).astype(np.float16)
    d0 = ((d1.astype(np.float32) @ W1) * (a0 > 0)).astype(np.float16)
    np.testing.assert_allclose(bb[0].astype(np.float32), d1.astype(np.float32), atol=2e-3)
    np.testing.assert_allclose(bb[1].astype(np.float32), d0.astype(np.float32), atol=2e-3)
    dW0 = d0.astype(np.float32).T @ x.astype(np.float32)
    np.testing.assert_allclose(gw[:Hd * I].reshape(Hd, I).astype(np.float32), dW0, atol=2e-2, rtol=2e-3)
    dX = d0.astype(np.float32) @ W0
    np.testing.assert_allclose(gi.astype(np.float32), dX, atol=5e-3, rtol=2e-3)


def test_combine_select_tie_rule():
    dens = np.array([1.0, 2.0, 2.0, 0.0], np.float32)
    rgb = np.ones((4, 3), np.float32) * 9
    maxd = np.array([2.0, 1.0, 2.0, 0.0], np.float32)
    best = np.zeros((4, 3), np.float32)
    m, b = oracle.combine_select(dens, rgb, maxd, best)
    assert list(m) == [2.0, 2.0, 2.0, 0.0]
    assert list(b[:, 0]) == [0.0, 9.0, 0.0, 0.0]               # strict '>': ties keep the earlier object


# ---------------------------------------------------------------- density-grid maintenance (nerf/renderer.py:356-508)
def test_mark_untrained_grid_known_answers():
    """One camera at (0,0,-3) looking along +z (identity rotation): cells in front within the frustum are seen, a camera
    looking away sees nothing; the count is the number of cameras."""
    H, C, bound = 16, 1, 1
    pose = np.eye(4, dtype=np.float32)
    pose[2, 3] = -3.0
    intr = (100.0, 100.0, 50.0, 50.0)          # tan(half fov) = 0.5
    grid = np.ones((C, H ** 3), np.float32)
    g, cnt = oracle.mark_untrained_grid(pose[None], intr, bound, C, H, grid)
    assert (cnt == 1).all() and (g == 1).all()                     # the whole unit cube is inside a 53-degree frustum from z = -3
    back = pose.copy()
    back[:3, :3] = np.diag([1, 1, -1]).astype(np.float32)           # looking along -z
    g, cnt = oracle.mark_untrained_grid(back[None], intr, bound, C, H, grid)
    assert (cnt == 0).all() and (g == -1).all()
    g, cnt = oracle.mark_untrained_grid(np.stack([pose, pose, back]), intr, bound, C, H, grid)
    assert (cnt == 2).all()
    # a narrow camera (tan = 0.05) sees only the cells near its axis: |x| < 0.05 * (z + 3) + 2 * half_cell
    narrow = (1000.0, 1000.0, 50.0, 50.0)
    g, cnt = oracle.mark_untrained_grid(pose[None], narrow, bound, C, H, grid)
    coords = oracle.morton3D_invert(np.arange(H ** 3, dtype=np.int32)).astype(np.float64)
    w = (2 * coords / (H - 1) - 1) * (1 - 1 / H)
    lim = 0.05 * (w[:, 2] + 3) + 2.0 / H
    want = (np.abs(w[:, 0]) < lim) & (np.abs(w[:, 1]) < lim)
    assert (cnt[0] > 0).sum() == want.sum() and np.array_equal(cnt[0] > 0, want)


def test_grid_update_known_answers():
    H, C = 8, 2
    H3 = H ** 3
    grid = np.zeros((C, H3), np.float32)
    grid[0, 5] = 2.0
    grid[1, 7] = -1.0                                               # untrained: never updated
    idx = np.array([[5, 5, 9], [7, 3, 3]], np.int32)
    sig = np.array([1.0, 1.5, 4.0, 9.0, 0.25, 0.5], np.float32)
    g, bits, mean = oracle.grid_update_apply(grid, C, H, sig, idx, 2.0, 0.5, 100.0)
    assert g[0, 5] == 3.0                   # max(2.0 * 0.5, max(1.0, 1.5) * 2.0)
    assert g[0, 9] == 8.0 and g[1, 3] == 1.0 and g[1, 7] == -1.0
    assert abs(mean - (3.0 + 8.0 + 1.0) / (C * H3)) < 1e-9
    want = np.zeros(C * H3, np.uint8)
    want[[5, 9, H3 + 3]] = 1                # threshold = min(mean, 100) = mean ~ 0.0117
    assert np.array_equal(np.unpackbits(bits, bitorder="little"), want)
    # sampling: with one occupied cell every pick lands on it; the first half are the random cells
    rc = np.array([[[1, 2, 3], [0, 0, 0]], [[7, 7, 7], [1, 0, 0]]], np.int32)
    occ = np.zeros((C, H3), np.float32)
    occ[0, 100] = 1.0
    idx2, xyz = oracle.grid_update_sample(occ, C, H, 2, rc, np.array([[0.0, 0.999], [0.3, 0.7]], np.float32), np.full((C * 4, 3), 0.5, np.float32))
    assert list(idx2[0]) == [int(oracle.morton3D(rc[0])[0]), 0, 100, 100]
    assert list(idx2[1]) == [511, 1, 511, 1]                         # cascade 1 has no occupied cell: the random cells are repeated
    c = oracle.morton3D_invert(np.array([100], np.int32))[0].astype(np.float32)
    assert np.array_equal(xyz[2], (2 * c / (H - 1) - 1) * np.float32(1 - 1 / H))     # jitter 0.5 -> centre of the cell, cascade 0 scale


# ---------------------------------------------------------------- density-grid maintenance against the REFERENCE's own torch code
def _golden_sigma(x):
    """The analytic density of the grid-maintenance fixture (tests/golden/make_golden.py, poly_sigma): add / multiply / max only, the
    same bits on every IEEE-754 machine."""
    import torch
    x = torch.from_numpy(np.ascontiguousarray(x))
    c = torch.tensor([0.1, -0.05, 0.2])
    r2 = ((x - c) * (x - c)).sum(-1)
    a = torch.clamp(1.0 - r2 * 2.5, min=0.0)
    q = ((x + 0.4) * (x + 0.4)).sum(-1)
    b = torch.clamp(1.0 - q * 16.0, min=0.0)
    return (a * a * 40.0 + b * 3.0).numpy()


def test_grid_maintenance_against_reference_fixture(golden_dir):
    """tests/golden/grid_maintenance.npz = NeRFRenderer.mark_untrained_grid / update_extra_state of the reference (nerf/renderer.py:356-508)
    executed on the CPU at grid size 32 with the jitter pinned to the cell centres and the randint draws recorded. The oracle's
    restatement replays the same calls: this pins the restatement (and through the GPU parity tests the kernels) to the reference."""
    g = np.load(os.path.join(golden_dir, "grid_maintenance.npz"))
    H, C, bound = int(g["H"]), int(g["cascade"]), float(g["bound"])
    H3 = H ** 3
    # mark_untrained_grid: torch's CPU matmul may associate the 3-term dot product differently -> cells exactly on a frustum plane
    got, cnt = oracle.mark_untrained_grid(g["mark_poses"], tuple(float(v) for v in g["mark_intrinsics"]), bound, C, H, g["mark_grid_before"])
    ref = g["mark_grid_after"]
    assert ((got == -1) != (ref == -1)).mean() < 1e-4
    assert 0.02 < (ref == -1).mean() < 0.98
    # two full sweeps
    grid = g["sweep_grid_before"].copy()
    half = np.full((C * H3, 3), 0.5, np.float32)
    for it in range(2):
        xyz = oracle.grid_cells_xyz(C, H, bound, half)
        grid, bits, mean = oracle.grid_update_apply(grid, C, H, _golden_sigma(xyz), None, float(g["density_scale"]), float(g["decay"]), float(g["density_thresh"]))
        assert np.array_equal(grid, g[f"sweep{it}_grid"]), f"sweep {it}: density_grid"
        assert abs(mean - float(g[f"sweep{it}_mean"])) <= 2e-6 * max(1.0, abs(mean))
        assert np.array_equal(bits, g[f"sweep{it}_bits"]), f"sweep {it}: bitfield"
    assert (grid[0, ::9] == -1).all()
    # steady-state update: recorded random cells, picks k into the ascending occupied list expressed as u = (k + 0.5) / n_occ
    coords = g["steady_coords"].astype(np.int32)
    n_occ = g["steady_n_occ"].astype(np.float64)
    assert (n_occ > 0).all()
    u = ((g["steady_pick"].astype(np.float64) + 0.5) / n_occ[:, None]).astype(np.float32)
    N = coords.shape[1]
    idx, xyz = oracle.grid_update_sample(grid, C, H, bound, coords, u, np.full((C * 2 * N, 3), 0.5, np.float32))
    grid2, bits2, mean2 = oracle.grid_update_apply(grid, C, H, _golden_sigma(xyz), idx, float(g["density_scale"]), float(g["decay"]), float(g["density_thresh"]))
    assert np.array_equal(grid2, g["steady_grid"])
    assert abs(mean2 - float(g["steady_mean"])) <= 2e-6 * max(1.0, abs(mean2))
    assert np.array_equal(bits2, g["steady_bits"])


# ---------------------------------------------------------------- multi-object combine against COMBINED.py's own methods
def test_combine_matches_reference_fixture(golden_dir):
    """tests/golden/combined.npz = COMBINED.py's best_densities_and_colors_v3 looped over 4 objects (:592-618, ties at zero and exact ties
    between objects) followed by image_depth_generation for both backgrounds, run from the reference file. The oracle's select is
    bit-exact (first object wins ties), its composite agrees to 2e-6 (torch.cumprod vs a running product)."""
    g = np.load(os.path.join(golden_dir, "combined.npz"))
    dens, rgbs = g["densities"], g["rgbs"]                     # [K,1,N,T], [K,1,N,T,3]
    K, _, N, T = dens.shape
    max_d, best = dens[0, 0].copy(), rgbs[0, 0].copy()
    for k in range(1, K):
        max_d, best = oracle.combine_select(dens[k, 0], rgbs[k, 0], max_d, best)
    assert np.array_equal(max_d, g["max_densities"][0]) and np.array_equal(best.reshape(N, T, 3), g["max_rgbs"][0])
    # the serial select written as one pass over objects (what one all-reduce(MAX) of (sigma, -rank) keys computes): same winner
    winner = np.zeros((N, T), np.int64)
    cur = dens[0, 0].copy()
    for k in range(1, K):
        take = dens[k, 0] > cur
        winner[take] = k
        cur = np.maximum(cur, dens[k, 0])
    assert np.array_equal(np.take_along_axis(rgbs[:, 0], winner[None, ..., None].repeat(3, -1), 0)[0], g["max_rgbs"][0])
    n2, f2 = oracle.near_far_from_aabb(g["rays_o"], g["rays_d"], g["aabb"], float(g["min_near"]))
    assert np.array_equal(n2, g["nears"]) and np.array_equal(f2, g["fars"])
    for bg, val in (("white", 1.0), ("black", 0.0)):
        img, dep = oracle.composite_fixed_steps(max_d, best.reshape(N, T, 3), g["nears"], g["fars"], val)
        ok = np.isfinite(g[f"depth_{bg}"])
        np.testing.assert_allclose(img, g[f"image_{bg}"], atol=2e-6, rtol=0)
        np.testing.assert_allclose(dep[ok], g[f"depth_{bg}"][ok], atol=2e-6, rtol=0)
        assert np.array_equal(np.isnan(dep), np.isnan(g[f"depth_{bg}"]))
