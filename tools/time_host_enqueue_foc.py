"""Host enqueue time of the FOC object-conditioned network's fixed-step training step next to its GPU time (and the plain topology's)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda", 0)
poses, intr = bench.make_training_rays(dev, 1, 8, seed=0)
gen = torch.Generator().manual_seed(1)
batches = [bench.sample_batch(poses, intr, dev, gen) for _ in range(8)]
for name in ("plain", "foc"):
    if name == "plain":
        m = bench.build_model(1, dev, seed=0).train()
        step = lambda b: bench.train_step(m, opt, sc, *b)
    else:
        m = bench.build_foc_model(1, dev, seed=0).train()
        yolo = bench.foc_yolo_details(dev, bench.NUM_RAYS, 7)
        step = lambda b: bench.foc_train_step(m, opt, sc, *b, yolo)
    opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True)
    sc = torch.amp.GradScaler("cuda")
    for i in range(60):
        step(batches[i % 8])
    torch.cuda.synchronize()
    n = int(os.environ.get("STEPS", "40"))
    t0 = time.perf_counter()
    for i in range(n):
        step(batches[i % 8])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host enqueue {1000 * (t1 - t0) / n:.3f} ms/step, total {1000 * (t2 - t0) / n:.3f} ms/step", flush=True)
