"""Timing helper: encoder + sigma-MLP forward (inference) through the separate nodes and through focnerf_amd.field."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from focnerf_amd.field import hashgrid_mlp

m = bench.build_model(1, torch.device("cuda", 0)).eval()
for B in (2 ** 21, 2 ** 17, 2 ** 13):
    x = torch.rand(B, 3, device="cuda") * 2 - 1
    for mode in ("separate", "fused"):
        def run():
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                if mode == "fused":
                    return hashgrid_mlp(m.encoder, m.sigma_net, x, m.bound)
                return m.sigma_net.forward_padded(m.encoder(x, bound=m.bound))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            run()
        torch.cuda.synchronize()
        print(B, mode, f"{1000 * (time.perf_counter() - t0) / n:.3f} ms")
