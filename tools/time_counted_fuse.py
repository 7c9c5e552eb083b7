"""Counted / plain level-major forward on ray-ordered points for several batch sizes and scene bounds (FOC_GRID_FUSE_SMALL is read once
per process: run once per setting). usage: python tools/time_counted_fuse.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from focnerf_amd.backend import _gridencoder
from focnerf_amd import raymarching

dev = torch.device("cuda", 0)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return 1000 * s.elapsed_time(e) / n


for bound in (1, 2):
    m = bench.build_model(bound, dev, seed=0)
    enc = m.encoder
    table = enc.embeddings.detach().half().contiguous()
    poses, intr = bench.make_training_rays(dev, bound, 8, seed=0)
    gen = torch.Generator().manual_seed(1)
    ro, rd, _ = bench.sample_batch(poses, intr, dev, gen)
    ro, rd = ro.view(-1, 3), rd.view(-1, 3)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, m.aabb_train, m.min_near)
    for rays, steps in ((4096, 512), (4096, 128), (4090, 128), (4095, 512)):
        t = torch.linspace(0, 1, steps, device=dev)[None, :]
        z = nears[:rays, None] + (fars - nears)[:rays, None] * t
        x = ((ro[:rays, None, :] + rd[:rays, None, :] * z[..., None]).clamp(-bound, bound).view(-1, 3) + bound) / (2 * bound)
        x = x.contiguous()
        B = x.shape[0]
        out = torch.empty(16, B, 2, device=dev, dtype=torch.half)
        S, H = float(np.log2(enc.per_level_scale)), enc.base_resolution
        plain = timed(lambda: _gridencoder.grid_encode_forward(x, table, enc.offsets, out, B, 3, 2, 16, S, H, None, 0, False, 0))
        counted = timed(lambda: _gridencoder.grid_encode_forward_counted(x, table, enc.offsets, out, B, 3, 2, 16, S, H, 0, False, 0))
        alone = timed(lambda: _gridencoder.grid_encode_forward_counted(x, table, enc.offsets, out, B, 3, 2, 16, S, H, 0, False, 0, standalone=True))
        print(f"fuse={os.environ.get('FOC_GRID_FUSE_SMALL', '1')} bound {bound} rays {rays} x {steps} = {B}: plain {plain:.1f} us, counted (+scans) {counted:.1f} us, count kernel + plain {alone:.1f} us", flush=True)

# the occupancy-grid sampler's own points (configs[2]): marched through the analytic occupancy grid, budgeted slot list
m2 = bench.build_model(2, dev, cuda_ray=True, seed=0).train()
enc = m2.encoder
table = enc.embeddings.detach().half().contiguous()
poses, intr = bench.make_training_rays(dev, 2, 8, seed=0)
gen = torch.Generator().manual_seed(1)
ro, rd, _ = bench.sample_batch(poses, intr, dev, gen)
ro, rd = ro.view(-1, 3).contiguous(), rd.view(-1, 3).contiguous()
near, far = raymarching.near_far_from_aabb(ro, rd, m2.aabb_train, m2.min_near)
def variants(x0):
    yield "as marched", x0
    yield "shuffled", x0[torch.randperm(x0.shape[0], device=dev)].contiguous()
    yield "stretched to 0..1", ((x0 - 0.5) * 1.75 + 0.5).clamp(0, 1).contiguous()
    yield "cut to a multiple of 1024", x0[:x0.shape[0] // 1024 * 1024].contiguous()
    yield "cut to a multiple of 1024, + 256", x0[:x0.shape[0] // 1024 * 1024 - 768].contiguous()


S, H = float(np.log2(enc.per_level_scale)), enc.base_resolution
for budget in (-1, 527872):
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(ro, rd, m2.bound, m2.density_bitfield, m2.cascade, m2.grid_size, near, far, counter, budget,
                                                            True, 128, False, 0, 1024)
    x0 = ((xyzs + 2) / 4).contiguous()
    n = rays[:, 2].float()
    print(f"budget {budget}: samples per ray min {n.min().item():.0f} mean {n.mean().item():.1f} max {n.max().item():.0f}, rays with none {(n == 0).sum().item()}")
    for variant, x in variants(x0):
        if x is None:
            continue
        B = x.shape[0]
        out = torch.empty(16, B, 2, device=dev, dtype=torch.half)
        plain = timed(lambda: _gridencoder.grid_encode_forward(x, table, enc.offsets, out, B, 3, 2, 16, S, H, None, 0, False, 0))
        counted = timed(lambda: _gridencoder.grid_encode_forward_counted(x, table, enc.offsets, out, B, 3, 2, 16, S, H, 0, False, 0))
        alone = timed(lambda: _gridencoder.grid_encode_forward_counted(x, table, enc.offsets, out, B, 3, 2, 16, S, H, 0, False, 0, standalone=True))
        print(f"fuse={os.environ.get('FOC_GRID_FUSE_SMALL', '1')} occupancy points {variant}, budget {budget}: B {B}, x range {x.min().item():.3f}..{x.max().item():.3f}: "
              f"plain {plain:.1f} us, counted (+scans) {counted:.1f} us, count kernel + plain {alone:.1f} us", flush=True)
