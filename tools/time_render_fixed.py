"""Timing helper: configs[1] render (fixed-step, 4096-ray chunks) — run under rocprofv3 --kernel-trace --stats.
Prints per view: host enqueue time (loop returned, nothing waited for) and wall time to completion, min / median over the views."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from focnerf_amd import synthetic

dev = torch.device("cuda", 0)
m = bench.build_model(1, dev, seed=0).eval()
poses, intr = bench.make_training_rays(dev, 1, 8, seed=0)
rays = [synthetic.get_rays(poses[v:v + 1], intr, 800, 800) for v in range(8)]
tile = os.environ.get("TILE")                       # "8x8", "4x16", "16x4": every 64 consecutive rays = one pixel tile (experiment: locality of a 64-ray block)
if tile:
    th, tw = (int(v) for v in tile.split("x"))
    yy, xx = torch.meshgrid(torch.arange(800), torch.arange(800), indexing="ij")
    key = ((yy // th) * (800 // tw) + (xx // tw)) * (th * tw) + (yy % th) * tw + (xx % tw)
    perm = torch.argsort(key.reshape(-1)).to(dev)
    rays = [(o[:, perm].contiguous(), d[:, perm].contiguous()) for o, d in rays]
ro, rd = rays[0]
views = int(os.environ.get("VIEWS", "8"))
kw = dict(staged=True, max_ray_batch=int(os.environ.get("CHUNK", "4096")), num_steps=512, upsample_steps=0, perturb=False, fused=True)
if os.environ.get("FIELDS") == "0":
    kw["return_fields"] = False
enq, tot = [], []
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    m.render(ro, rd, **kw)
    torch.cuda.synchronize()
    for i in range(views):
        t0 = time.perf_counter()
        out = m.render(*rays[i % 8], **kw)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enq.append(t1 - t0); tot.append(t2 - t0)
        del out
print(f"enqueue min {min(enq):.4f} med {np.median(enq):.4f}  s/view min {min(tot):.4f} mean {np.mean(tot):.4f} max {max(tot):.4f} (views cycle through 8 poses)")
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(views):
        out = m.render(*rays[i % 8], **kw)
    torch.cuda.synchronize()
print(f"back to back (no synchronisation between views): s/view {(time.perf_counter() - t0) / views:.4f}")
print("s/view", float(np.mean(tot)))
