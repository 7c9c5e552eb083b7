"""Timing helper: NeRFRenderer.update_extra_state (density-grid maintenance, every 16 training steps in the reference trainer)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda", 0)
for bound in (1, 2):
    m = bench.build_model(bound, dev, cuda_ray=True, seed=0).train()
    with torch.autocast("cuda", dtype=torch.float16):
        for _ in range(2):
            m.update_extra_state()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            m.update_extra_state()
        torch.cuda.synchronize()
    print("bound", bound, "cascade", m.cascade, "update_extra_state ms", 1000 * (time.perf_counter() - t0) / n)
