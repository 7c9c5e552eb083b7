"""render_field4 over an 800 x 800 view in 4096-ray chunks (what every rank does per view in the combined render): host time to enqueue the
view and wall time to its completion. Run on the GPU box."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from focnerf_amd import synthetic
from focnerf_amd.field import half_cache_scope
from focnerf_amd.fixedstep import render_field4
from focnerf_amd.rayorder import view_tiling

dev = torch.device("cuda", 0)
m = bench.build_foc_model(1, dev, seed=0).eval()
yolo = bench.foc_yolo_details(dev, 1, 50)
poses, intr = bench.make_training_rays(dev, 1, 8, seed=0)
o, d = synthetic.get_rays(poses[:1], intr, 800, 800)
o, d = o[0].contiguous(), d[0].contiguous()
if os.environ.get("TILES", "1") != "0":
    p = view_tiling(d)
    o, d = o[p], d[p]
n, chunk = o.shape[0], 4096
buf = torch.empty(chunk, 512, 4, dtype=torch.float32, device=dev)
with torch.no_grad(), half_cache_scope():
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for lo in range(0, n, chunk):
            hi = min(lo + chunk, n)
            render_field4(m, o[lo:hi], d[lo:hi], num_steps=512, yolo_details=yolo, out=buf[: hi - lo])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"enqueue {1e3 * (t1 - t0):.2f} ms, complete {1e3 * (t2 - t0):.2f} ms per view", flush=True)
