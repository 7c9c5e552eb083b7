"""Fixed-step training step for several ray counts / samples per ray: GPU time per sample should not depend on how the batch size
divides by 256 / 1024 (looks for launch-geometry accidents like an even spacing of the counting workgroups)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda", 0)
model = bench.build_model(1, dev, seed=0).train()
opt = torch.optim.Adam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True)
scaler = torch.amp.GradScaler("cuda")
poses, intr = bench.make_training_rays(dev, 1, 8, seed=0)
gen = torch.Generator().manual_seed(1)
for rays, steps in ((4096, 512), (4095, 512), (4000, 512), (3333, 512), (4096, 500), (4096, 384), (2048, 512), (2047, 512), (1024, 512), (1000, 512)):
    bench.NUM_RAYS, bench.NUM_STEPS = rays, steps
    batches = [bench.sample_batch(poses, intr, dev, gen) for _ in range(4)]
    for i in range(6):
        bench.train_step(model, opt, scaler, *batches[i % 4])
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for i in range(12):
        bench.train_step(model, opt, scaler, *batches[i % 4])
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 12
    print(f"rays {rays} x {steps} = {rays * steps}: {ms:.3f} ms/step, {1e6 * ms / (rays * steps):.3f} ns/sample", flush=True)
