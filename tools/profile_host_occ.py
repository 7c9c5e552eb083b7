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
pr = cProfile.Profile()
pr.enable()
for i in range(100):
    bench.cuda_ray_train_step(m2, opt2, sc2, *b2[i % 4])
pr.disable()
torch.cuda.synchronize()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(45)
print(out.getvalue()[:9000])
