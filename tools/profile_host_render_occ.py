"""cProfile of the Python side of the occupancy-grid render loop (march_rays / composite_rays, host-bound): where the enqueue time goes."""
import cProfile, pstats, os, sys, io, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from focnerf_amd import synthetic

dev = torch.device("cuda", 0)
m2 = bench.build_model(2, dev, cuda_ray=True, seed=0).eval()
poses2, intr = bench.make_training_rays(dev, 2, 8, seed=0)
ro2, rd2 = synthetic.get_rays(poses2[:1], intr, 800, 800)
kw = dict(staged=False, perturb=False, dt_gamma=1 / 128, max_steps=1024, T_thresh=1e-4, device_compaction=True)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    m2.render(ro2, rd2, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m2.render(ro2, rd2, **kw)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"enqueue {t1 - t0:.4f} s, total {time.perf_counter() - t0:.4f} s per view")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        m2.render(ro2, rd2, **kw)
    pr.disable()
    torch.cuda.synchronize()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(40)
print(out.getvalue()[:9000])
