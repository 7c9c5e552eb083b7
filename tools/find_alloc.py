"""Which call makes the one device allocation inside the timed region of bench.py? (run on the GPU box)"""
import os, sys, pickle
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
model = bench.build_model(1, dev, seed=0).train()
opt = torch.optim.Adam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True)
sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: 0.1 ** min(it / 30000, 1))
scaler = torch.amp.GradScaler("cuda")
poses, intr = bench.make_training_rays(dev, 1, 8, seed=0)
gen = torch.Generator().manual_seed(1000)
batches = [bench.sample_batch(poses, intr, dev, gen) for _ in range(64)]
for i in range(300):
    bench.train_step(model, opt, scaler, *batches[i % 64], fused=True, sched=sched)
torch.cuda.synchronize()
a0 = torch.cuda.memory_stats(dev)["num_device_alloc"]
torch.cuda.memory._record_memory_history(max_entries=200000)
for i in range(200):
    bench.train_step(model, opt, scaler, *batches[i % 64], fused=True, sched=sched)
torch.cuda.synchronize()
print("device allocs in 200 steps:", torch.cuda.memory_stats(dev)["num_device_alloc"] - a0)
snap = torch.cuda.memory._snapshot()
for tr in snap["device_traces"]:
    for ev in tr:
        if ev["action"] in ("segment_alloc", "segment_free"):
            fr = [f"{f['filename'].split('/')[-1]}:{f['line']}:{f['name']}" for f in ev.get("frames", [])[:14]]
            print(ev["action"], ev["size"], " <- ".join(fr))
