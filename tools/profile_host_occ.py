"""cProfile of the Python side of the occupancy-grid training step (which is host-bound): where the enqueue time goes."""
import cProfile, pstats, os, sys, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda", 0)
m2 = bench.build_model(2, dev, cuda_ray=True, seed=0).train()
opt2 = torch.optim.Adam(m2.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True)
sc2 = torch.amp.GradScaler("cuda")
poses2, intr = bench.make_training_rays(dev, 2, 8, seed=0)
gen = torch.Generator().manual_seed(1)
b2 = [bench.sample_batch(poses2, intr, dev, gen) for _ in range(4)]
for i in range(40):
    bench.cuda_ray_train_step(m2, opt2, sc2, *b2[i % 4])
    if i == 15:
        m2.mean_count = int(m2.step_counter[:16, 0].sum().item() / 16)
torch.cuda.synchronize()
import time
# host time per step without the profiler: the loop's wall time until the LAST step is enqueued (the queue is drained first, and 100 steps of
# ~0.8 ms do not fill it), against the time until the GPU is done
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(100):
    bench.cuda_ray_train_step(m2, opt2, sc2, *b2[i % 4])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"100 steps: host enqueue {10 * (t1 - t0):.3f} ms/step, until the GPU is done {10 * (t2 - t0):.3f} ms/step")
# (over 100 steps a host that is faster than the GPU runs into the launch queue's depth and is throttled to the GPU's pace less a constant)
# the host's OWN cost: bursts of 8 steps (~500 launches, inside the queue) from an idle GPU
burst = []
for rep in range(12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(8):
        bench.cuda_ray_train_step(m2, opt2, sc2, *b2[i % 4])
    burst.append((time.perf_counter() - t0) / 8)
    torch.cuda.synchronize()
burst.sort()
print(f"bursts of 8 steps from an idle GPU: host enqueue min {1e3 * burst[0]:.3f} median {1e3 * burst[len(burst) // 2]:.3f} ms/step")
# the node as one library call each way against the call-by-call node, interleaved in THIS process (separate processes differ by more than the effect)
res = {"1": [], "0": []}
for rep in range(20):
    for mode in ("1", "0"):
        os.environ["FOC_OCC_NATIVE_NODE"] = mode
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(8):
            bench.cuda_ray_train_step(m2, opt2, sc2, *b2[i % 4])
        res[mode].append((time.perf_counter() - t0) / 8)
        torch.cuda.synchronize()
for mode, name in (("1", "one call each way"), ("0", "call by call")):
    v = sorted(res[mode])
    print(f"interleaved, {name}: host enqueue min {1e3 * v[0]:.3f} median {1e3 * v[len(v) // 2]:.3f} ms/step")
os.environ.pop("FOC_OCC_NATIVE_NODE", None)
pr = cProfile.Profile()
pr.enable()
for i in range(100):
    bench.cuda_ray_train_step(m2, opt2, sc2, *b2[i % 4])
pr.disable()
torch.cuda.synchronize()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(45)
print(out.getvalue()[:9000])
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(35)
print(out.getvalue()[:7000])
