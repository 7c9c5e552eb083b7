"""Timing helper: configs[1] render (fixed-step, 4096-ray chunks) — run under rocprofv3 --kernel-trace --stats."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from focnerf_amd import synthetic

dev = torch.device("cuda", 0)
m = bench.build_model(1, dev, seed=0).eval()
poses, intr = bench.make_training_rays(dev, 1, 8, seed=0)
ro, rd = synthetic.get_rays(poses[:1], intr, 800, 800)
kw = dict(staged=True, max_ray_batch=4096, num_steps=512, upsample_steps=0, perturb=False, fused=True)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    m.render(ro, rd, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        m.render(ro, rd, **kw)
    torch.cuda.synchronize()
print("s/view", (time.perf_counter() - t0) / 3)
