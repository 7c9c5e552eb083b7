"""GPU parity: multi-object combine kernels (csrc/combine.hip) vs the CPU oracle — bit-exact select (strict '>' tie rule),
key pack/unpack round trip, and the single-rank ObjectCombiner against the serial COMBINED.py loop."""
import numpy as np
import pytest
import torch

import oracle
from util import to_np

pytestmark = pytest.mark.gpu


def _fields(K, N, T, seed):
    rng = np.random.default_rng(seed)
    dens = (rng.random((K, N, T)) ** 4 * 40).astype(np.float32)
    dens[rng.random((K, N, T)) < 0.5] = 0
    if K > 1:
        dens[1, :, :8] = dens[0, :, :8]
    rgb = rng.random((K, N, T, 3)).astype(np.float32)
    nears = (rng.random(N) * 0.5 + 0.2).astype(np.float32)
    fars = nears + (rng.random(N) * 2 + 0.5).astype(np.float32)
    return dens, rgb, nears, fars


def test_serial_combine_matches_oracle_bit_exact():
    from focnerf_amd.combine import combine_serial, composite_fixed_steps
    K, N, T = 4, 300, 128
    dens, rgb, nears, fars = _fields(K, N, T, 3)
    md, best = combine_serial([(torch.from_numpy(dens[k]).cuda(), torch.from_numpy(rgb[k]).cuda()) for k in range(K)])
    m_ref, b_ref = dens[0].copy(), rgb[0].copy()
    for k in range(1, K):
        m_ref, b_ref = oracle.combine_select(dens[k], rgb[k], m_ref, b_ref)
    assert np.array_equal(to_np(md), m_ref.reshape(N, T)) and np.array_equal(to_np(best), b_ref.reshape(N, T, 3))
    # independent statement of COMBINED.py:247-251 in torch
    md_t, b_t = torch.from_numpy(dens[0]), torch.from_numpy(rgb[0])
    for k in range(1, K):
        d, c = torch.from_numpy(dens[k]), torch.from_numpy(rgb[k])
        b_t = torch.where(d[..., None] > md_t[..., None], c, b_t)
        md_t = torch.maximum(d, md_t)
    assert torch.equal(md.cpu(), md_t) and torch.equal(best.cpu(), b_t)
    img4, depth = composite_fixed_steps(md, best, torch.from_numpy(nears).cuda(), torch.from_numpy(fars).cuda(), 1.0)
    # the device kernel evaluates torch's DEVICE linspace (fused multiply-add in the upper half), the CPU oracle the CPU one: <= 1 ulp in z
    i_ref, d_ref = oracle.composite_fixed_steps(m_ref.reshape(N, T), b_ref.reshape(N, T, 3), nears, fars, 1.0, clamp01=True)
    np.testing.assert_allclose(to_np(img4), i_ref, atol=1e-4)
    np.testing.assert_allclose(to_np(depth), d_ref, atol=1e-4)


def test_key_pack_unpack_and_single_rank_combiner():
    from focnerf_amd.combine import HipCombineOps, ObjectCombiner
    N, T = 257, 64
    dens, rgb, nears, fars = _fields(1, N, T, 5)
    d = torch.from_numpy(dens[0]).cuda()
    c = torch.from_numpy(rgb[0]).cuda()
    for rank in (0, 3, 7):
        keys = HipCombineOps.pack_keys(d, rank)
        k = to_np(keys)
        assert np.array_equal((k >> 32).astype(np.uint32).view(np.float32), dens[0])
        assert np.all((k & 0xFFFFFFFF) == 0xFFFFFFFF - rank)
        md, masked = HipCombineOps.unpack(keys, rank, c)
        assert torch.equal(md, d) and torch.equal(masked, c)
        md2, masked2 = HipCombineOps.unpack(keys, rank + 1, c)
        assert torch.equal(md2, d) and torch.all(masked2 == 0)
    # ordering property the MAX all-reduce relies on: larger sigma wins, equal sigma -> lower rank wins
    a = HipCombineOps.pack_keys(torch.tensor([1.0, 2.0, 2.0], device="cuda"), 1)
    b = HipCombineOps.pack_keys(torch.tensor([2.0, 1.0, 2.0], device="cuda"), 0)
    assert (a[0] < b[0]) and (a[1] > b[1]) and (a[2] < b[2])
    comb = ObjectCombiner(rank=0, world_size=1)
    img, dep = comb.render_chunk(d, c, torch.from_numpy(nears).cuda(), torch.from_numpy(fars).cuda(), bg=1.0)
    i_ref, d_ref = oracle.composite_fixed_steps(dens[0], rgb[0], nears, fars, 1.0, clamp01=True)
    np.testing.assert_allclose(to_np(img), i_ref, atol=1e-4)
    np.testing.assert_allclose(to_np(dep), d_ref, atol=1e-4)


def test_combine_kernels_replay_the_reference_fixture():
    """tests/golden/combined.npz (COMBINED.py's own best_densities_and_colors_v3 loop + image_depth_generation) through the HIP path:
    the serial select, the key form used across ranks (simulated here: K packs, elementwise max, unpack per rank, sum), the composite."""
    import os
    from focnerf_amd.combine import combine_serial, composite_fixed_steps, HipCombineOps
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "combined.npz"))
    dens, rgbs = g["densities"][:, 0], g["rgbs"][:, 0]          # [K,N,T], [K,N,T,3]
    K, N, T = dens.shape
    fields = [(torch.from_numpy(dens[k]).cuda(), torch.from_numpy(rgbs[k]).cuda()) for k in range(K)]
    md, best = combine_serial(fields)
    assert np.array_equal(to_np(md), g["max_densities"][0]) and np.array_equal(to_np(best), g["max_rgbs"][0])
    # what ObjectCombiner.select does with K ranks: all-reduce(MAX) of the keys, all-reduce(SUM) of the winner-masked colours
    keys = torch.stack([HipCombineOps.pack_keys(fields[k][0], k) for k in range(K)]).max(dim=0).values
    md2, summed = None, 0
    for k in range(K):
        md2, masked = HipCombineOps.unpack(keys, k, fields[k][1])
        summed = summed + masked
    assert np.array_equal(to_np(md2), g["max_densities"][0]) and np.array_equal(to_np(summed), g["max_rgbs"][0])
    nears, fars = torch.from_numpy(g["nears"]).cuda(), torch.from_numpy(g["fars"]).cuda()
    for bg, val in (("white", 1.0), ("black", 0.0)):
        img4, depth = composite_fixed_steps(md, best, nears, fars, val)
        ok = np.isfinite(g[f"depth_{bg}"])
        np.testing.assert_allclose(to_np(img4), g[f"image_{bg}"], atol=1e-4, rtol=0)       # RGB / alpha within the 1e-4 target
        np.testing.assert_allclose(to_np(depth)[ok], g[f"depth_{bg}"][ok], atol=1e-4, rtol=0)
