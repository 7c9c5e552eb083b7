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


# ---------------------------------------------------------------------------------------------- packed fields, fused select + composite
def _pack(dens, rgb):
    return torch.from_numpy(np.concatenate([dens[..., None], rgb], -1).astype(np.float32)).cuda().contiguous()


@pytest.mark.parametrize("K,N,T", [(1, 33, 64), (2, 257, 512), (8, 100, 200), (16, 40, 65), (19, 21, 70)])
def test_fused_select_composite_equals_select_then_composite(K, N, T):
    """foc_combine_select_composite == foc_combine_select (object loop) followed by foc_composite_fixed_steps, bit for bit: merged
    field (incl. the tie rule and more objects than one call's 16), both backgrounds, depth; T not a multiple of 64; 1 <= K <= 19."""
    from focnerf_amd.combine import combine_packed, combine_serial, composite_fixed_steps
    dens, rgb, nears, fars = _fields(K, N, T, 10 + K)
    if K > 2:
        dens[K - 1, :, 8:16] = dens[2, :, 8:16]
    nr, fr = torch.from_numpy(nears).cuda(), torch.from_numpy(fars).cuda()
    md, best = combine_serial([(torch.from_numpy(dens[k]).cuda(), torch.from_numpy(rgb[k]).cuda()) for k in range(K)])
    img4, depth, merged = combine_packed([_pack(dens[k], rgb[k]) for k in range(K)], nr, fr, (1.0, 0.0), want_merged=True)
    assert torch.equal(merged[..., 0], md) and torch.equal(merged[..., 1:], best)
    for q, bg in enumerate((1.0, 0.0)):
        i_ref, d_ref = composite_fixed_steps(md, best, nr, fr, bg)
        assert torch.equal(img4[q], i_ref) and torch.equal(depth, d_ref)
    i2, d2 = combine_packed([_pack(dens[k], rgb[k]) for k in range(K)], nr, fr, (0.25,))
    assert i2.shape == (1, N, 4) and torch.equal(d2, depth)


def test_fused_select_propagates_nan_like_torch_maximum():
    from focnerf_amd.combine import combine_packed
    N, T = 4, 64
    a = torch.rand(N, T, 4, device="cuda")
    b = torch.rand(N, T, 4, device="cuda")
    a[0, 3, 0] = float("nan")
    b[1, 5, 0] = float("nan")
    nr, fr = torch.full((N,), 0.2, device="cuda"), torch.full((N,), 2.0, device="cuda")
    _, _, merged = combine_packed([a, b], nr, fr, (1.0,), want_merged=True)
    md = torch.maximum(b[..., 0], a[..., 0])
    best = torch.where((b[..., 0] > a[..., 0])[..., None], b[..., 1:], a[..., 1:])
    assert torch.equal(torch.isnan(merged[..., 0]), torch.isnan(md)) and torch.isnan(merged[0, 3, 0]) and torch.isnan(merged[1, 5, 0])
    ok = ~torch.isnan(md)
    assert torch.equal(merged[..., 0][ok], md[ok]) and torch.equal(merged[..., 1:], best)


def test_configs4_editable_fixture_on_the_device():
    """BASELINE configs[4]: editable.py's own 8-object loop with the edited-object offset, two views (tests/golden/editable.npz, made by
    the reference's methods). On the device: (1) the select over the eight packed fields reproduces the reference's max_densities /
    max_rgbs bit for bit, the composites both backgrounds within 1e-4; (2) the object-side pack kernel (own compositing weights ->
    mask w > 1e-10 -> packed field) reproduces what the reference's `run` returned from the raw per-sample field along the object's
    own — edited — rays."""
    import os
    from focnerf_amd._lib import lib, ptr, stream_of, check
    from focnerf_amd.combine import RayEditor, combine_packed, object_type_from_ckpt
    from focnerf_amd import raymarching
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "editable.npz"))
    K, T = int(fx["K"]), int(fx["T"])
    aabb = torch.from_numpy(fx["aabb"]).cuda()
    editor = RayEditor(str(fx["edit_object"]), tuple(fx["offset"]))
    for v in range(2):
        o, d = torch.from_numpy(fx[f"v{v}_rays_o"]).cuda(), torch.from_numpy(fx[f"v{v}_rays_d"]).cuda()
        N = o.shape[0]
        nears, fars = raymarching.near_far_from_aabb(o, d, aabb, float(fx["min_near"]))
        assert np.array_equal(to_np(nears), fx[f"v{v}_nears"]) and np.array_equal(to_np(fars), fx[f"v{v}_fars"])
        fields = []
        for k in range(K):
            mo, md_ = editor(o, d, object_type_from_ckpt(str(fx["ckpts"][k])))
            assert np.array_equal(to_np(mo), fx[f"v{v}_mod_o"][k])
            np.testing.assert_allclose(to_np(md_), fx[f"v{v}_mod_d"][k], atol=1e-7, rtol=0)      # F.normalize on the device vs the CPU
            n_k, f_k = raymarching.near_far_from_aabb(mo.contiguous(), torch.from_numpy(fx[f"v{v}_mod_d"][k]).cuda(), aabb, float(fx["min_near"]))
            assert np.array_equal(to_np(n_k), fx[f"v{v}_own_near_far"][k, 0]) and np.array_equal(to_np(f_k), fx[f"v{v}_own_near_far"][k, 1])
            sig = torch.from_numpy(fx[f"v{v}_densities"][k]).cuda().reshape(-1).contiguous()
            raw = torch.from_numpy(fx[f"v{v}_raw_rgbs"][k]).cuda().reshape(-1, 3).contiguous()
            f4 = torch.empty(N, T, 4, device="cuda")
            check(lib.foc_fixed_field_pack(ptr(sig), ptr(raw), ptr(n_k), ptr(f_k), None, None, 1.0, N, T, 1.0, 1e-10, None, None, None, ptr(f4),
                                           0, stream_of(sig)), "fixed_field_pack")
            got = to_np(f4)
            assert np.array_equal(got[..., 0], fx[f"v{v}_densities"][k])
            want = fx[f"v{v}_rgbs"][k]
            # a sample is either kept (raw colour) or zeroed; the decision w > 1e-10 may differ only where w is within rounding of 1e-10
            kept_ref, kept_got = (want != 0).any(-1), (got[..., 1:] != 0).any(-1)
            flips = kept_ref != kept_got
            assert flips.mean() < 2e-3, f"object {k}: {flips.sum()} mask decisions differ"
            same = ~flips
            assert np.array_equal(got[..., 1:][same], want[same])
            fields.append(_pack(fx[f"v{v}_densities"][k], want))                   # the reference's own fields feed the select
        img4, depth, merged = combine_packed(fields, nears, fars, (1.0, 0.0), want_merged=True)
        assert np.array_equal(to_np(merged[..., 0]), fx[f"v{v}_max_densities"]) and np.array_equal(to_np(merged[..., 1:]), fx[f"v{v}_max_rgbs"])
        ok = np.isfinite(fx[f"v{v}_depth_white"])
        np.testing.assert_allclose(to_np(img4[0]), fx[f"v{v}_image_white"], atol=1e-4, rtol=0)
        np.testing.assert_allclose(to_np(img4[1]), fx[f"v{v}_image_black"], atol=1e-4, rtol=0)
        np.testing.assert_allclose(to_np(depth)[ok], fx[f"v{v}_depth_white"][ok], atol=1e-4, rtol=0)


def _object(seed, bound=1):
    from focnerf_amd.network import NeRFNetwork
    torch.manual_seed(seed)
    m = NeRFNetwork(bound=bound).cuda().eval()
    m.encoder.embeddings.data.uniform_(-0.5, 0.5)
    return m


def test_render_field4_equals_run_return_fields():
    """The fused object-side producer (sample -> encoder -> whole-field kernel -> own weights, mask, pack) against the general path:
    `run(..., return_fields=True)` (densities, rgbs) packed with torch ops. Same kernels upstream, so bit for bit."""
    from focnerf_amd import synthetic
    from focnerf_amd.combine import pack_field4
    from focnerf_amd.fixedstep import render_field4
    m = _object(3)
    rays_o, rays_d = synthetic.make_view_rays(24, 24, 1, 1, seed=5, device="cuda")
    o, d = rays_o[0, :300].contiguous(), rays_d[0, :300].contiguous()
    T = 96
    f4 = render_field4(m, o, d, num_steps=T)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        ref = m.run(o[None], d[None], num_steps=T, upsample_steps=0, perturb=False, fused=True, return_fields=True)
    want = pack_field4(ref["densities"], ref["rgbs"])
    assert f4.shape == (300, T, 4) and torch.equal(f4, want)
    assert (f4[..., 0] > 0).any() and (f4[..., 1:] != 0).any()
    buf = torch.zeros(300, T, 4, device="cuda")
    assert render_field4(m, o, d, num_steps=T, out=buf) is buf and torch.equal(buf, want)


def test_resident_objects_through_the_combiner_equal_the_reload_loop(tmp_path):
    """SURVEY.md §8f-4 end to end (COMBINED.py:592-618 with nerf/utils.py:1431-1530 checkpoints). K per-object checkpoints in the reference
    trainer's layout — numpy scalars in `stats`, optimizer / ema entries, best checkpoints without density_grid — are (a) loaded ONCE
    with `load_objects`, kept resident and rendered through render_field4 -> ObjectCombiner.render_view; (b) rendered the reference's
    way: ONE network object, `load_checkpoint` per object per view, run(return_fields=True) over the view's chunks into whole-view
    [N,T] / [N,T,3] tensors, best_densities_and_colors_v3 loop, image_depth_generation per background. Same image, bit for bit."""
    import numpy as np
    from focnerf_amd import raymarching, synthetic
    from focnerf_amd.checkpoint import load_checkpoint, load_objects
    from focnerf_amd.combine import ObjectCombiner, RayEditor, combine_serial, composite_fixed_steps
    from focnerf_amd.fixedstep import render_field4
    from focnerf_amd.network import NeRFNetwork
    K, T, chunk = 3, 64, 128
    paths = []
    for k in range(K):
        m = _object(20 + k)
        state = {"epoch": 5, "global_step": 500, "stats": {"loss": [0.1], "valid_loss": [0.1], "results": [np.float64(30.0 + k)], "checkpoints": [],
                                                            "best_result": np.float64(30.0 + k)},
                 "model": {n: t.detach().cpu() for n, t in m.state_dict().items()}}
        if k == 1:
            state["optimizer"] = {"state": {}, "param_groups": [{"lr": 0.01, "params": [0, 1, 2]}]}
            state["ema"] = {"decay": 0.95, "num_updates": 3, "shadow_params": [p.detach().cpu() for p in m.parameters()], "collected_params": None}
        p = tmp_path / f"bottle_{k}.pth" if k == 1 else tmp_path / f"obj{k}.pth"
        torch.save(state, str(p))
        paths.append(str(p))
    rays_o, rays_d = synthetic.make_view_rays(20, 20, 1, 1, seed=9, device="cuda")
    o, d = rays_o[0].contiguous(), rays_d[0].contiguous()          # 400 rays: chunks of 128, 128, 128, 16
    N = o.shape[0]
    editor_a = RayEditor("bottle", (0.02, -0.01, 0.05), freeze_first_view=False)
    editor_b = RayEditor("bottle", (0.02, -0.01, 0.05), freeze_first_view=False)
    from focnerf_amd.combine import object_type_from_ckpt

    # (a) resident objects
    models = load_objects(paths, lambda: NeRFNetwork(bound=1), torch.device("cuda"))
    nears, fars = raymarching.near_far_from_aabb(o, d, models[0].aabb_infer, models[0].min_near)
    fns = []
    for k, mk in enumerate(models):
        mo, md = editor_a(o, d, object_type_from_ckpt(paths[k]))
        fns.append(lambda lo, hi, out, mk=mk, mo=mo, md=md: render_field4(mk, mo[lo:hi], md[lo:hi], num_steps=T, out=out))
    comb = ObjectCombiner(rank=0, world_size=1)
    img_a, dep_a = comb.render_view(fns, N, nears, fars, T, bgs=(1.0, 0.0), max_ray_batch=chunk)

    # (b) the reference's flow
    one = NeRFNetwork(bound=1).cuda().eval()
    fields = []
    for k in range(K):
        load_checkpoint(one, paths[k], map_location="cuda")
        mo, md = editor_b(o, d, object_type_from_ckpt(paths[k]))
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            res = one.render(mo[None], md[None], staged=True, max_ray_batch=chunk, num_steps=T, upsample_steps=0, perturb=False, fused=True, return_fields=True)
        fields.append((res["densities"][0].clone(), res["rgbs"][0].clone()))
    md_, best = combine_serial(fields)
    for q, bg in enumerate((1.0, 0.0)):
        i_ref, d_ref = composite_fixed_steps(md_, best, nears, fars, bg)
        assert torch.equal(img_a[q], i_ref), f"background {bg}"
        assert torch.equal(dep_a, d_ref)
    assert not torch.equal(fields[0][0], fields[1][0])               # the objects do differ


def test_combine_at_baseline_size_walked_in_chunks():
    """configs[3] at BASELINE size on one device: an 800 x 800 view x 512 samples, K = 4 resident objects (the bench's
    `resident_4_objects_one_gpu` objects and view), walked in 4096-ray chunks by ObjectCombiner.render_view. Properties that do not need
    an oracle of this size: (1) every chunk's rows of the view's image / depth are `combine_packed` of that chunk's four fields, bit for
    bit (checked on a spread of chunks incl. the ragged last one); (2) an object that ties with an earlier one on EVERY sample — same
    densities, other colours — never wins a sample (COMBINED.py:247-251: strict `>`, the first checkpoint keeps ties); (3) the image
    checksum is the one bench.py reports for the leg."""
    import bench
    from focnerf_amd import raymarching as rm, synthetic
    from focnerf_amd.combine import ObjectCombiner, combine_packed
    from focnerf_amd.field import half_cache_scope
    from focnerf_amd.rayorder import view_tiling
    dev = torch.device("cuda", 0)
    poses, intr = bench.make_training_rays(dev, 1, 8, seed=0)
    rays_o, rays_d = synthetic.get_rays(poses[:1], intr, bench.VIEW, bench.VIEW)
    vo, vd = rays_o[0].contiguous(), rays_d[0].contiguous()
    order = view_tiling(vd)
    vo, vd = vo.index_select(0, order), vd.index_select(0, order)             # the bench walks the view in 8 x 8 pixel tiles
    N, T, chunk = bench.VIEW * bench.VIEW, bench.NUM_STEPS, 4096
    fns = bench.resident_object_fields(dev, 1, vo, vd, 4)
    with torch.no_grad(), half_cache_scope():
        model0 = bench.build_model(1, dev, seed=0)
        vn, vf = rm.near_far_from_aabb(vo, vd, model0.aabb_infer, model0.min_near)
        comb = ObjectCombiner(rank=0, world_size=1)
        img4, depth = comb.render_view(fns, N, vn, vf, T, max_ray_batch=chunk)
        assert img4.shape == (2, N, 4) and depth.shape == (N,)
        n_chunks = -(-N // chunk)
        for c in (0, 1, n_chunks // 2, n_chunks - 2, n_chunks - 1):
            lo, hi = c * chunk, min((c + 1) * chunk, N)
            fields = [fn(lo, hi, None).view(hi - lo, T, 4) for fn in fns]
            i_ref, d_ref = combine_packed(fields, vn[lo:hi], vf[lo:hi], (1.0, 0.0))
            assert torch.equal(img4[:, lo:hi], i_ref) and torch.equal(depth[lo:hi].nan_to_num(), d_ref.nan_to_num()), f"chunk {c}"
        assert (img4[0, :, :3] < 0.99).any() and img4[0, :, 3].max() > 0.5
        checksum = float(img4.double().sum().item())
        # (2) ties: object 2 := object 0's densities with the colours inverted — it must not change a single value anywhere
        def tied(lo, hi, out):
            f = fns[0](lo, hi, out)
            f.view(-1, 4)[:, 1:].mul_(-1).add_(1)
            return f
        img4_t, depth_t = comb.render_view([fns[0], fns[1], tied, fns[3]], N, vn, vf, T, max_ray_batch=chunk)
        ref4, refd = comb.render_view([fns[0], fns[1], fns[3]], N, vn, vf, T, max_ray_batch=chunk)
        assert torch.equal(img4_t, ref4) and torch.equal(depth_t.nan_to_num(), refd.nan_to_num()), "a later object tying on every sample must never win one"
        lo, hi = 200 * chunk // 2, 200 * chunk // 2 + chunk
        f0, ft = fns[0](lo, hi, None).clone(), tied(lo, hi, None).clone()
        _, _, merged = combine_packed([f0, ft], vn[lo:hi], vf[lo:hi], (1.0,), want_merged=True)
        assert torch.equal(merged, f0) and not torch.equal(ft[..., 1:], f0[..., 1:])
        # (3) what bench.py reports for the same objects and view
        img4_b, _ = ObjectCombiner(rank=0, world_size=1).render_view(bench.resident_object_fields(dev, 1, vo, vd, 4), N, vn, vf, T, max_ray_batch=chunk)
        assert float(img4_b.double().sum().item()) == checksum
