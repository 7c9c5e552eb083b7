"""How long the host takes to enqueue one headline training step (no synchronisation inside), next to the GPU time of the step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda", 0)
model = bench.build_model(1, dev, seed=0).train()
opt = torch.optim.Adam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True)
scaler = torch.amp.GradScaler("cuda")
poses, intr = bench.make_training_rays(dev, 1, 8, seed=0)
gen = torch.Generator().manual_seed(1)
batches = [bench.sample_batch(poses, intr, dev, gen) for _ in range(4)]
for i in range(10):
    bench.train_step(model, opt, scaler, *batches[i % 4], fused=True)
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for i in range(n):
    bench.train_step(model, opt, scaler, *batches[i % 4], fused=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1000 * (t1 - t0) / n:.3f} ms/step, total {1000 * (t2 - t0) / n:.3f} ms/step, cpus {os.cpu_count()}")
st0 = torch.cuda.memory_stats()
for i in range(20):
    bench.train_step(model, opt, scaler, *batches[i % 4], fused=True)
torch.cuda.synchronize()
st1 = torch.cuda.memory_stats()
for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "segment.all.allocated", "segment.all.freed", "allocation.all.allocated",
          "reserved_bytes.all.current", "reserved_bytes.all.peak", "allocated_bytes.all.peak", "num_sync_all_streams"):
    print(k, st0.get(k), "->", st1.get(k))
