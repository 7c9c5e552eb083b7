"""The headline training step (bench.py: fixed-step path, 512 samples per ray, fused tail, Adam + GradScaler) at other batch sizes than the
benchmark's 4096 rays: ms per step and samples/s per rays-per-step, one line each and one JSON line. Same model, optimiser and step function as
bench.py; only the number of rays drawn per step changes."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

dev = torch.device("cuda", 0)
out = []
for rays in [int(v) for v in (sys.argv[1:] or ["512", "1024", "2048", "4096", "8192", "16384"])]:
    bench.NUM_RAYS = rays
    model = bench.build_model(1, dev, cuda_ray=False, seed=0).train()
    opt = torch.optim.Adam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True)
    scaler = torch.amp.GradScaler("cuda")
    poses, intr = bench.make_training_rays(dev, 1, 8, seed=0)
    gen = torch.Generator().manual_seed(1000)
    batches = [bench.sample_batch(poses, intr, dev, gen) for _ in range(8)]
    assert batches[0][0].shape[-2] == rays, batches[0][0].shape
    for i in range(12):
        bench.train_step(model, opt, scaler, *batches[i % 8])
    torch.cuda.synchronize()
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(20):
            bench.train_step(model, opt, scaler, *batches[i % 8])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        best = dt if best is None else min(best, dt)
    row = {"rays_per_step": rays, "samples_per_step": rays * bench.NUM_STEPS, "ms_per_step": 1e3 * best, "samples_per_sec": rays * bench.NUM_STEPS / best}
    out.append(row)
    print(f"{rays:6d} rays/step  {row['ms_per_step']:7.3f} ms/step  {row['samples_per_sec'] / 1e9:6.3f} G samples/s", flush=True)
    del model, opt, scaler, batches
    torch.cuda.empty_cache()
print(json.dumps(out))
