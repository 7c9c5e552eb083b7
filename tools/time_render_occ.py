"""Timing helper: configs[2] render (march_rays / composite_rays loop) — run under rocprofv3 --kernel-trace --stats."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from focnerf_amd import synthetic

dev = torch.device("cuda", 0)
m2 = bench.build_model(2, dev, cuda_ray=True, seed=0).eval()
poses2, intr = bench.make_training_rays(dev, 2, 8, seed=0)
ro2, rd2 = synthetic.get_rays(poses2[:1], intr, 800, 800)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    m2.render(ro2, rd2, staged=False, perturb=False, dt_gamma=1 / 128, max_steps=1024, T_thresh=1e-4)      # (first use of a view this size: the tile permutation is made)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        m2.render(ro2, rd2, staged=False, perturb=False, dt_gamma=1 / 128, max_steps=1024, T_thresh=1e-4, device_compaction=True)
    torch.cuda.synchronize()
print("s/view", (time.perf_counter() - t0) / 3)
