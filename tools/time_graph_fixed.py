"""Headline fixed-step training step: eager vs replayed as one HIP graph."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from focnerf_amd.graph import GraphedStep

dev = torch.device("cuda", 0)
model = bench.build_model(1, dev, seed=0).train()
opt = torch.optim.Adam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True, capturable=True)
scaler = torch.amp.GradScaler("cuda")
poses, intr = bench.make_training_rays(dev, 1, 8, seed=0)
gen = torch.Generator().manual_seed(1)
batches = [bench.sample_batch(poses, intr, dev, gen) for _ in range(4)]
for i in range(100):
    bench.train_step(model, opt, scaler, *batches[i % 4], fused=True)


def timed(fn, n=40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(*batches[i % 4])
    torch.cuda.synchronize()
    return 1000 * (time.perf_counter() - t0) / n


eager = timed(lambda o, d, t: bench.train_step(model, opt, scaler, o, d, t, fused=True))
step = GraphedStep(lambda o, d, t: bench.train_step(model, opt, scaler, o, d, t, fused=True), batches[0])
graphed = timed(step)
print(f"eager {eager:.3f} ms/step, graphed {graphed:.3f} ms/step")
