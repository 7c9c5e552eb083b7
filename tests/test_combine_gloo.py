"""CPU, world_size 2 over gloo: the host / collective logic of the one-object-per-rank combine
(focnerf_amd/combine.py). The device kernels are replaced by CPU ops DEFINED HERE (backed by the
oracle) and injected through the `ops` parameter — the product itself has no CPU path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle


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


def _fields(K, N, T, seed):
    rng = np.random.default_rng(seed)
    dens = (rng.random((K, N, T)) ** 4 * 40).astype(np.float32)
    dens[rng.random((K, N, T)) < 0.5] = 0          # plenty of exact ties at 0
    if K > 1:
        dens[1, :, :8] = dens[0, :, :8]            # and exact non-zero ties: the lower rank must win
    rgb = rng.random((K, N, T, 3)).astype(np.float32)
    nears = (rng.random(N) * 0.5 + 0.2).astype(np.float32)
    fars = nears + (rng.random(N) * 2 + 0.5).astype(np.float32)
    return dens, rgb, nears, fars


def _worker(rank, world, port, N, T, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
    # serial reference: COMBINED.py's loop over checkpoints with the strict-'>' select, then one composite
    from focnerf_amd.combine import combine_serial
    md_s, best_s = combine_serial([(torch.from_numpy(dens[k]), torch.from_numpy(rgb[k])) for k in range(world)], ops=CpuOps)
    img_s, dep_s = oracle.composite_fixed_steps(md_s.numpy(), best_s.numpy(), nears, fars, 1.0, clamp01=True)
    for r in range(world):
        g = np.load(os.path.join(tmp_path, f"r{r}.npz"))
        assert np.array_equal(g["md"], md_s.numpy()), "merged density must be bit-exact"
        assert np.array_equal(g["best"], best_s.numpy()), "merged colour must be bit-exact (incl. tie rule)"
        assert np.array_equal(g["img"], img_s) and np.array_equal(g["dep"], dep_s)
        assert np.all(g["fi"] == 3.0) and np.all(g["fw"] == 0.5)       # fast mode: plain sums over ranks


def test_single_rank_is_identity():
    from focnerf_amd.combine import ObjectCombiner
    dens, rgb, nears, fars = _fields(1, 10, 16, 1)
    comb = ObjectCombiner(rank=0, world_size=1, ops=CpuOps)
    md, best = comb.select(torch.from_numpy(dens[0]), torch.from_numpy(rgb[0]))
    assert np.array_equal(md.numpy(), dens[0]) and np.array_equal(best.numpy(), rgb[0])


def test_modify_rays_for_object():
    from focnerf_amd.combine import modify_rays_for_object
    o = torch.zeros(5, 3)
    assert torch.equal(modify_rays_for_object(o, (0.01, 0.01, 0.6)), torch.tensor([[0.01, 0.01, 0.6]]).expand(5, 3))
