"""Where the samples of the occupancy-grid render loop (legacy/nerf/renderer.py:323-372) go: per iteration the number of live rays, the burst
length, the samples evaluated (live x burst, padded to the marching alignment), how many of them are real (dt > 0), and how many the compositing
actually uses (samples in front of the point where the ray's transmittance falls under T_thresh). Run on the GPU box."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from focnerf_amd import synthetic, raymarching
from focnerf_amd.renderer import _MARCH_ALIGN

dev = torch.device("cuda", 0)
m = bench.build_model(2, dev, cuda_ray=True, seed=0).eval()
poses, intr = bench.make_training_rays(dev, 2, 8, seed=0)
o, d = synthetic.get_rays(poses[:1], intr, 800, 800)
o, d = o.reshape(-1, 3).contiguous(), d.reshape(-1, 3).contiguous()
n = o.shape[0]
print("occupied cells", float((m.density_grid > min(m.mean_density, m.density_thresh)).float().mean()), "mean density", float(m.mean_density))
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    near, far = raymarching.near_far_from_aabb(o, d, m._aabb(), m.min_near)
    opacity, depth, image = (torch.zeros(n, *t, dtype=torch.float32, device=dev) for t in ((), (), (3,)))
    alive = torch.arange(n, dtype=torch.int32, device=dev)
    t_now = near.clone()
    marched, it, tot_eval, tot_real = 0, 0, 0, 0
    while marched < 1024 and alive.shape[0] > 0:
        live = alive.shape[0]
        burst = max(min(n // live, 8), 1)
        xyzs, dirs, deltas = raymarching.march_rays(live, burst, alive, t_now, o, d, m.bound, m.density_bitfield, m.cascade, m.grid_size, near, far,
                                                    _MARCH_ALIGN, False, 1 / 128, 1024)
        real = int((deltas[:, 0] > 0).sum())
        sig, rgb = m(xyzs, dirs)
        raymarching.composite_rays(live, burst, alive, t_now, sig, rgb, deltas, opacity, depth, image, 1e-4)
        alive = alive[alive >= 0]
        tot_eval += xyzs.shape[0]
        tot_real += real
        if it < 12 or it % 10 == 0 or alive.shape[0] < 2000:
            print(f"it {it:4d} live {live:7d} burst {burst} evaluated {xyzs.shape[0]:7d} real {real:7d} ({real / max(xyzs.shape[0], 1):.2f}) -> live {alive.shape[0]}")
        marched += burst
        it += 1
print("iterations", it, "samples evaluated", tot_eval, "real", tot_real, "per ray", tot_eval / n)
