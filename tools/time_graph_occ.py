"""Occupancy-grid training step: eager vs replayed as one HIP graph."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from focnerf_amd.graph import GraphedStep

dev = torch.device("cuda", 0)
m2 = bench.build_model(2, dev, cuda_ray=True, seed=0).train()
opt2 = torch.optim.Adam(m2.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True, capturable=True)
sc2 = torch.amp.GradScaler("cuda")
poses2, intr = bench.make_training_rays(dev, 2, 8, seed=0)
gen = torch.Generator().manual_seed(1)
b2 = [bench.sample_batch(poses2, intr, dev, gen) for _ in range(4)]
for i in range(40):
    bench.cuda_ray_train_step(m2, opt2, sc2, *b2[i % 4])
    if i == 15:
        m2.mean_count = int(m2.step_counter[:16, 0].sum().item() / 16)
torch.cuda.synchronize()


def timed(fn, n=40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(*b2[i % 4])
    torch.cuda.synchronize()
    return 1000 * (time.perf_counter() - t0) / n


eager = timed(lambda o, d, t: bench.cuda_ray_train_step(m2, opt2, sc2, o, d, t))
step = GraphedStep(lambda o, d, t: bench.cuda_ray_train_step(m2, opt2, sc2, o, d, t), b2[0])
graphed = timed(step)
l0 = float(step(*b2[0]))
for i in range(50):
    step(*b2[i % 4])
l1 = float(step(*b2[0]))
print(f"eager {eager:.3f} ms/step, graphed {graphed:.3f} ms/step, loss {l0:.5f} -> {l1:.5f}, samples/step {float(m2.step_counter[:, 0].float().max()):.0f}")
