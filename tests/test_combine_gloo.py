"""CPU, world_size 2, 4 and 8 over gloo: the host / collective logic of the one-object-per-rank combine
(focnerf_amd/combine.py). The device kernels are replaced by CPU ops DEFINED HERE (backed by the
oracle) and injected through the `ops` parameter — the product itself has no CPU path.

Covered: the by-ray exchange (`render_view`: all-to-all of packed fields -> select + composite on the owner of each ray slice ->
all-gather) against the serial object loop, bit for bit, with ragged chunks, more objects than ranks, overlap on and off; the
reference's own editable.py fixture (8 objects, edited-object ray offset, two views) replayed through 4 ranks x 2 objects; the
round-1 key/all-reduce form."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class CpuOps:
    @staticmethod
    def select(dens, rgb, max_dens, best_rgb):
        m, b = oracle.combine_select(dens.numpy(), rgb.numpy(), max_dens.numpy(), best_rgb.numpy())
        max_dens.copy_(torch.from_numpy(m)); best_rgb.copy_(torch.from_numpy(b))

    @staticmethod
    def pack_keys(dens, rank):
        d = np.maximum(dens.numpy(), 0).astype(np.float32)
        d[~(dens.numpy() > 0)] = 0
        keys = (d.view(np.uint32).astype(np.int64) << 32) | np.int64(0xFFFFFFFF - rank)
        return torch.from_numpy(keys)

    @staticmethod
    def unpack(keys, rank, rgb):
        k = keys.numpy()
        mine = (k & 0xFFFFFFFF) == (0xFFFFFFFF - rank)
        max_dens = (k >> 32).astype(np.uint32).view(np.float32)
        masked = np.where(mine[..., None], rgb.numpy(), 0).astype(np.float32)
        return torch.from_numpy(max_dens.copy()), torch.from_numpy(masked)

    @staticmethod
    def composite(sigmas, rgbs, nears, fars, bg):
        i4, d = oracle.composite_fixed_steps(sigmas.numpy(), rgbs.numpy(), nears.numpy(), fars.numpy(), bg, clamp01=True)
        return torch.from_numpy(i4), torch.from_numpy(d)

    @staticmethod
    def select4(field4, acc4):
        f, a = field4.numpy(), acc4.numpy()
        m, b = oracle.combine_select(np.ascontiguousarray(f[..., 0]), np.ascontiguousarray(f[..., 1:]), np.ascontiguousarray(a[..., 0]),
                                     np.ascontiguousarray(a[..., 1:]))
        acc4.copy_(torch.from_numpy(np.concatenate([m[..., None], b], -1)))

    @staticmethod
    def select_composite(fields4, nears, fars, bgs, want_merged=False):
        acc = fields4[0].clone().contiguous()
        for f in fields4[1:]:
            CpuOps.select4(f.contiguous(), acc)
        a = acc.numpy()
        imgs, dep = [], None
        for bg in bgs:
            i4, dep = oracle.composite_fixed_steps(np.ascontiguousarray(a[..., 0]), np.ascontiguousarray(a[..., 1:]), nears.numpy(), fars.numpy(), float(bg),
                                                   clamp01=True)
            imgs.append(i4)
        out = (torch.from_numpy(np.stack(imgs)), torch.from_numpy(dep))
        return out + (acc,) if want_merged else out


def _fields(K, N, T, seed):
    rng = np.random.default_rng(seed)
    dens = (rng.random((K, N, T)) ** 4 * 40).astype(np.float32)
    dens[rng.random((K, N, T)) < 0.5] = 0          # plenty of exact ties at 0
    if K > 1:
        dens[1, :, :8] = dens[0, :, :8]            # and exact non-zero ties: the lower rank must win
    if K > 3:
        dens[3, :, 8:12] = dens[2, :, 8:12]        # a tie inside one rank's run of objects (2 objects per rank at world 2)
    rgb = rng.random((K, N, T, 3)).astype(np.float32)
    nears = (rng.random(N) * 0.5 + 0.2).astype(np.float32)
    fars = nears + (rng.random(N) * 2 + 0.5).astype(np.float32)
    return dens, rgb, nears, fars


def _packed(dens, rgb):
    return torch.from_numpy(np.concatenate([dens[..., None], rgb], -1).astype(np.float32))


def _serial(dens, rgb, nears, fars, bgs):
    """COMBINED.py's loop over checkpoints with the strict-'>' select, then one composite per background — on the oracle."""
    from focnerf_amd.combine import combine_serial
    K = dens.shape[0]
    md, best = combine_serial([(torch.from_numpy(dens[k]), torch.from_numpy(rgb[k])) for k in range(K)], ops=CpuOps)
    imgs, dep = [], None
    for bg in bgs:
        i4, dep = oracle.composite_fixed_steps(md.numpy(), best.numpy(), nears, fars, float(bg), clamp01=True)
        imgs.append(i4)
    return md.numpy(), best.numpy(), np.stack(imgs), dep


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _worker(rank, world, port, N, T, out_dir):
    _init(rank, world, port)
    from focnerf_amd.combine import ObjectCombiner
    dens, rgb, nears, fars = _fields(world, N, T, 0)
    comb = ObjectCombiner(ops=CpuOps)
    assert comb.rank == rank and comb.world == world
    md, best = comb.select(torch.from_numpy(dens[rank]), torch.from_numpy(rgb[rank]))
    img, dep = comb.render_chunk(torch.from_numpy(dens[rank]), torch.from_numpy(rgb[rank]), torch.from_numpy(nears), torch.from_numpy(fars), bg=1.0)
    fi, fd, fw = comb.render_chunk_fast(torch.full((N, 3), float(rank + 1)), torch.full((N,), 0.5), torch.full((N,), 0.25))
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), md=md.numpy(), best=best.numpy(), img=img.numpy(), dep=dep.numpy(), fi=fi.numpy(), fw=fw.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("N", [37, 64])       # ragged and even ray counts over 2 ranks
def test_two_rank_combine_equals_serial(tmp_path, N):
    world, T = 2, 48
    mp.spawn(_worker, args=(world, _free_port(), N, T, str(tmp_path)), nprocs=world, join=True)
    dens, rgb, nears, fars = _fields(world, N, T, 0)
    md_s, best_s, img_s, dep_s = _serial(dens, rgb, nears, fars, (1.0,))
    for r in range(world):
        g = np.load(os.path.join(tmp_path, f"r{r}.npz"))
        assert np.array_equal(g["md"], md_s), "merged density must be bit-exact"
        assert np.array_equal(g["best"], best_s), "merged colour must be bit-exact (incl. tie rule)"
        assert np.array_equal(g["img"], img_s[0]) and np.array_equal(g["dep"], dep_s)
        assert np.all(g["fi"] == 3.0) and np.all(g["fw"] == 0.5)       # fast mode: plain sums over ranks


# ---------------------------------------------------------------------------------------------- exchange by ray
def _view_worker(rank, world, port, K, N, T, chunk, overlap, out_dir):
    _init(rank, world, port)
    from focnerf_amd.combine import ObjectCombiner
    dens, rgb, nears, fars = _fields(K, N, T, 3)
    per_rank = K // world
    mine = range(rank * per_rank, (rank + 1) * per_rank)          # consecutive runs of the checkpoint list
    calls = []

    def make_fn(k):
        f4 = _packed(dens[k], rgb[k])

        def fn(lo, hi, out):
            calls.append((k, lo, hi, out is not None))
            if out is not None and k % 2 == 0:                     # both protocols: write into the offered buffer, or return a fresh tensor
                out.copy_(f4[lo:hi])
                return out
            return f4[lo:hi].clone()
        return fn
    comb = ObjectCombiner(ops=CpuOps)
    img, dep = comb.render_view([make_fn(k) for k in mine], N, torch.from_numpy(nears), torch.from_numpy(fars), T, bgs=(1.0, 0.0), max_ray_batch=chunk,
                                overlap=overlap)
    n_chunks = (N + chunk - 1) // chunk
    assert len(calls) == per_rank * n_chunks
    np.savez(os.path.join(out_dir, f"v{rank}.npz"), img=img.numpy(), dep=dep.numpy(), sent=np.int64(comb.bytes_sent))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,K,N,chunk,overlap", [
    (2, 2, 96, 32, True),        # even everything
    (2, 2, 101, 32, True),       # ragged last chunk (5 rays: slices of 3 and 2), odd slice
    (2, 4, 70, 33, False),       # two objects per rank (local pre-merge incl. a tie inside the run), odd chunk, no overlap
    (4, 4, 50, 16, True),        # world 4; last chunk of 2 rays: two ranks own nothing of it
    (4, 8, 45, 45, True),        # configs[4]-shaped: 8 objects on 4 ranks, a single ragged chunk
    (8, 8, 43, 43, True),        # configs[4] at its real world size: 8 objects on 8 ranks, a ray count that 8 does not divide
    (8, 8, 69, 32, True),        # ... and a last chunk of 5 rays: three of the eight ranks own no ray of it
    (8, 8, 69, 32, False),       # the same without overlap
    (8, 16, 21, 8, True),        # two objects per rank at world 8, a last chunk of 5
])
def test_render_view_by_ray_exchange_equals_serial(tmp_path, world, K, N, chunk, overlap):
    T = 24
    mp.spawn(_view_worker, args=(world, _free_port(), K, N, T, chunk, overlap, str(tmp_path)), nprocs=world, join=True)
    dens, rgb, nears, fars = _fields(K, N, T, 3)
    _, _, img_s, dep_s = _serial(dens, rgb, nears, fars, (1.0, 0.0))
    for r in range(world):
        g = np.load(os.path.join(tmp_path, f"v{r}.npz"))
        assert g["img"].shape == (2, N, 4) and g["dep"].shape == (N,)
        assert np.array_equal(g["img"], img_s), f"rank {r}: image differs from the serial object loop"
        assert np.array_equal(g["dep"], dep_s)
        # (p-1)/p of 16 B per sample leave every rank (plus the padded slices of ragged chunks and the final gather)
        assert g["sent"] >= (world - 1) * N * T * 16 // world


def _one_rank_worker(rank, world, port, K, N, T, chunk, out_dir):
    _init(rank, world, port)
    from focnerf_amd.combine import ObjectCombiner
    dens, rgb, nears, fars = _fields(K, N, T, 3)
    fns = [lambda lo, hi, out, f4=_packed(dens[k], rgb[k]): f4[lo:hi].clone() for k in range(K)]
    res = {}
    for name, comb in (("plain", ObjectCombiner(ops=CpuOps)), ("forced", ObjectCombiner(ops=CpuOps, collectives_at_world_1=True))):
        assert comb.world == 1 and comb.xch == (name == "forced")
        for overlap in (True, False):
            img, dep = comb.render_view(fns, N, torch.from_numpy(nears), torch.from_numpy(fars), T, bgs=(1.0, 0.0), max_ray_batch=chunk, overlap=overlap)
            res[f"{name}_img_{int(overlap)}"], res[f"{name}_dep_{int(overlap)}"] = img.numpy(), dep.numpy()
        md, best = comb.select(torch.from_numpy(dens[0]), torch.from_numpy(rgb[0]))
        i4, dp = comb.render_chunk(torch.from_numpy(dens[0]), torch.from_numpy(rgb[0]), torch.from_numpy(nears), torch.from_numpy(fars), bg=1.0)
        fi, fd, fw = comb.render_chunk_fast(torch.full((N, 3), 2.0), torch.full((N,), 0.5), torch.full((N,), 0.25))
        res.update({f"{name}_md": md.numpy(), f"{name}_best": best.numpy(), f"{name}_i4": i4.numpy(), f"{name}_dp": dp.numpy(), f"{name}_fi": fi.numpy(),
                    f"{name}_fw": fw.numpy()})
        assert comb.bytes_sent == 0                                   # nothing leaves a single rank, collectives or not
    np.savez(os.path.join(out_dir, "one.npz"), **res)
    dist.destroy_process_group()


@pytest.mark.parametrize("K,N,chunk", [(3, 101, 32), (2, 40, 64)])       # ragged last chunk; one chunk larger than the view
def test_single_rank_issuing_its_collectives_equals_the_exchange_free_path(tmp_path, K, N, chunk):
    """`ObjectCombiner(collectives_at_world_1=True)`: the switch tests/test_gpu_rccl.py uses to push the N > 1 code through RCCL on a one-GPU
    box. Here over gloo with one rank: every entry point gives what the exchange-free combiner gives and what the serial object loop gives."""
    T = 24
    mp.spawn(_one_rank_worker, args=(1, _free_port(), K, N, T, chunk, str(tmp_path)), nprocs=1, join=True)
    g = np.load(os.path.join(tmp_path, "one.npz"))
    dens, rgb, nears, fars = _fields(K, N, T, 3)
    _, _, img_s, dep_s = _serial(dens, rgb, nears, fars, (1.0, 0.0))
    for ov in (0, 1):
        for name in ("plain", "forced"):
            assert np.array_equal(g[f"{name}_img_{ov}"], img_s) and np.array_equal(g[f"{name}_dep_{ov}"], dep_s), (name, ov)
    for key in ("md", "best", "i4", "dp", "fi", "fw"):
        assert np.array_equal(g[f"plain_{key}"], g[f"forced_{key}"]), key


def _editable_worker(rank, world, port, out_dir):
    """editable.npz through 4 ranks x 2 objects: rays per object from RayEditor (state carried over the two views), the fixture's own
    per-object fields as what each object's evaluation returned."""
    _init(rank, world, port)
    from focnerf_amd.combine import ObjectCombiner, RayEditor, object_type_from_ckpt
    fx = np.load(os.path.join(GOLDEN, "editable.npz"))
    K, T, chunk = int(fx["K"]), int(fx["T"]), int(fx["chunk"])
    per_rank = K // world
    mine = list(range(rank * per_rank, (rank + 1) * per_rank))
    editor = RayEditor(str(fx["edit_object"]), tuple(fx["offset"]))
    comb = ObjectCombiner(ops=CpuOps)
    res = {}
    for v in range(2):
        o, d = torch.from_numpy(fx[f"v{v}_rays_o"]), torch.from_numpy(fx[f"v{v}_rays_d"])
        fns = []
        for k in mine:
            mo, md = editor(o, d, object_type_from_ckpt(str(fx["ckpts"][k])))
            assert np.array_equal(mo.numpy(), fx[f"v{v}_mod_o"][k]) and np.array_equal(md.numpy(), fx[f"v{v}_mod_d"][k]), (v, k)
            f4 = _packed(fx[f"v{v}_densities"][k], fx[f"v{v}_rgbs"][k])
            fns.append(lambda lo, hi, out, f4=f4: f4[lo:hi].clone())
        img, dep = comb.render_view(fns, o.shape[0], torch.from_numpy(fx[f"v{v}_nears"]), torch.from_numpy(fx[f"v{v}_fars"]), T, bgs=(1.0, 0.0),
                                    max_ray_batch=chunk)
        res[f"img{v}"], res[f"dep{v}"] = img.numpy(), dep.numpy()
    np.savez(os.path.join(out_dir, f"e{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_configs4_editable_fixture_over_four_and_eight_ranks(tmp_path, world):
    """BASELINE configs[4] (8 objects, editable.py offset render) at world 4 (two objects per rank) and at its REAL world size 8 (one object per
    rank) on the reference's own numbers: the images the reference's methods produced (tests/golden/make_golden.py editable_fixture) come out of
    the sharded path within the oracle-vs-torch summation tolerance, on every rank, for both views and both backgrounds."""
    mp.spawn(_editable_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    fx = np.load(os.path.join(GOLDEN, "editable.npz"))
    for r in range(world):
        g = np.load(os.path.join(tmp_path, f"e{r}.npz"))
        for v in range(2):
            np.testing.assert_allclose(g[f"img{v}"][0], fx[f"v{v}_image_white"], atol=3e-6, rtol=0)
            np.testing.assert_allclose(g[f"img{v}"][1], fx[f"v{v}_image_black"], atol=3e-6, rtol=0)
            np.testing.assert_allclose(g[f"dep{v}"], fx[f"v{v}_depth_white"], atol=3e-6, rtol=0)


def test_editable_fixture_serial_select_is_bit_exact():
    """The reference's merged field of configs[4] (max_densities / max_rgbs after its 8-object loop) from the packed select, bit for bit,
    and RayEditor / object_type_from_ckpt against what editable.py returned — including the rays of the first view that the reference
    keeps handing to the unedited objects on every later view."""
    from focnerf_amd.combine import RayEditor, combine_packed, object_type_from_ckpt
    fx = np.load(os.path.join(GOLDEN, "editable.npz"))
    K = int(fx["K"])
    assert [str(object_type_from_ckpt(str(c))) for c in fx["ckpts"]] == [str(t) for t in fx["object_types"]]
    editor = RayEditor(str(fx["edit_object"]), tuple(fx["offset"]))
    for v in range(2):
        o, d = torch.from_numpy(fx[f"v{v}_rays_o"]), torch.from_numpy(fx[f"v{v}_rays_d"])
        for k in range(K):
            mo, md = editor(o, d, object_type_from_ckpt(str(fx["ckpts"][k])))
            assert np.array_equal(mo.numpy(), fx[f"v{v}_mod_o"][k]) and np.array_equal(md.numpy(), fx[f"v{v}_mod_d"][k])
        fields = [_packed(fx[f"v{v}_densities"][k], fx[f"v{v}_rgbs"][k]) for k in range(K)]
        img, dep, merged = combine_packed(fields, torch.from_numpy(fx[f"v{v}_nears"]), torch.from_numpy(fx[f"v{v}_fars"]), (1.0, 0.0), want_merged=True,
                                          ops=CpuOps)
        assert np.array_equal(merged.numpy()[..., 0], fx[f"v{v}_max_densities"])
        assert np.array_equal(merged.numpy()[..., 1:], fx[f"v{v}_max_rgbs"])
        np.testing.assert_allclose(img.numpy()[0], fx[f"v{v}_image_white"], atol=3e-6, rtol=0)
        np.testing.assert_allclose(img.numpy()[1], fx[f"v{v}_image_black"], atol=3e-6, rtol=0)
    # the unedited objects of view 1 were given view 0's rays (editable.py:465-471); without the freeze they get the view's own
    assert np.array_equal(fx["v1_mod_o"][0], fx["v0_rays_o"]) and not np.array_equal(fx["v1_rays_o"], fx["v0_rays_o"])
    free = RayEditor(str(fx["edit_object"]), tuple(fx["offset"]), freeze_first_view=False)
    o1, d1 = torch.from_numpy(fx["v1_rays_o"]), torch.from_numpy(fx["v1_rays_d"])
    assert free(o1, d1, "book")[0] is o1


def test_single_rank_is_identity():
    from focnerf_amd.combine import ObjectCombiner
    dens, rgb, nears, fars = _fields(1, 10, 16, 1)
    comb = ObjectCombiner(rank=0, world_size=1, ops=CpuOps)
    md, best = comb.select(torch.from_numpy(dens[0]), torch.from_numpy(rgb[0]))
    assert np.array_equal(md.numpy(), dens[0]) and np.array_equal(best.numpy(), rgb[0])
    # render_view without a process group: K resident objects on one device
    K, N, T = 3, 29, 16
    dens, rgb, nears, fars = _fields(K, N, T, 2)
    fns = [lambda lo, hi, out, k=k: _packed(dens[k], rgb[k])[lo:hi].clone() for k in range(K)]
    img, dep = comb.render_view(fns, N, torch.from_numpy(nears), torch.from_numpy(fars), T, bgs=(1.0, 0.0), max_ray_batch=8)
    _, _, img_s, dep_s = _serial(dens, rgb, nears, fars, (1.0, 0.0))
    assert np.array_equal(img.numpy(), img_s) and np.array_equal(dep.numpy(), dep_s)
    assert comb.bytes_sent == 0


def test_modify_rays_for_object():
    from focnerf_amd.combine import modify_rays_for_object
    o = torch.zeros(5, 3)
    assert torch.equal(modify_rays_for_object(o, (0.01, 0.01, 0.6)), torch.tensor([[0.01, 0.01, 0.6]]).expand(5, 3))
    d = torch.tensor([[0.0, 3.0, 4.0]]).expand(5, 3)
    o2, d2 = modify_rays_for_object(o, (0.01, 0.01, 0.6), d)
    assert torch.equal(o2, modify_rays_for_object(o, (0.01, 0.01, 0.6))) and torch.allclose(d2, torch.tensor([[0.0, 0.6, 0.8]]).expand(5, 3))
