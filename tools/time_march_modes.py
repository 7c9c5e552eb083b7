"""march_rays_train: wave-per-ray vs lane-per-ray walk for several ray counts (FOC_MARCH_SERIAL is read once per process)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from focnerf_amd import raymarching

dev = torch.device("cuda", 0)
m2 = bench.build_model(2, dev, cuda_ray=True, seed=0).train()
poses, intr = bench.make_training_rays(dev, 2, 8, seed=0)
gen = torch.Generator().manual_seed(1)
for n in (1024, 4096, 8192, 12288, 16384, 32768, 65536):
    bench.NUM_RAYS = n
    ro, rd, _ = bench.sample_batch(poses, intr, dev, gen)
    ro, rd = ro.view(-1, 3).contiguous(), rd.view(-1, 3).contiguous()
    near, far = raymarching.near_far_from_aabb(ro, rd, m2.aabb_train, m2.min_near)
    budget = 129 * n
    def run():
        counter = torch.zeros(2, dtype=torch.int32, device=dev)
        raymarching.march_rays_train(ro, rd, m2.bound, m2.density_bitfield, m2.cascade, m2.grid_size, near, far, counter, budget, True, 128, False, 1 / 128, 1024)
    for _ in range(3):
        run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        run()
    e.record(); torch.cuda.synchronize()
    print(f"FOC_MARCH_SERIAL={os.environ.get('FOC_MARCH_SERIAL', 'auto')} rays {n}: {100 * s.elapsed_time(e):.1f} us per call (march + scan + emit + torch allocs)", flush=True)
