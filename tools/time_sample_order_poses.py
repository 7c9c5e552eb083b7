"""Is the gain of the 64-ray block order a property of one camera pose?  Eight poses, rows 400..405 of each view, ray-major vs 64-ray blocks;
also the same chunk taken down image COLUMNS (neighbouring rays differ along the camera's up vector instead of its right vector)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from focnerf_amd import synthetic, raymarching
from focnerf_amd.backend import _gridencoder
from focnerf_amd.field import _half_of

dev = torch.device("cuda", 0)
m = bench.build_model(1, dev, seed=0)
poses, intr = bench.make_training_rays(dev, 1, 8, seed=0)
T, R = 512, 4096
enc = m.encoder
emb = _half_of(enc.embeddings)
L = enc.offsets.shape[0] - 1


def time_fwd(pts):
    pts = pts.reshape(-1, 3).contiguous()
    M = pts.shape[0]
    planes = torch.empty(L, M, 2, device=dev, dtype=torch.half)
    args = (pts, emb, enc.offsets, planes, M, 3, 2, L, float(np.log2(enc.per_level_scale)), enc.base_resolution, None, enc.gridtype_id, enc.align_corners,
            enc.interp_id)
    for _ in range(3):
        _gridencoder.grid_encode_forward(*args)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        _gridencoder.grid_encode_forward(*args)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 10


for v in range(poses.shape[0]):
    ro, rd = synthetic.get_rays(poses[v:v + 1], intr, bench.VIEW, bench.VIEW)
    ro, rd = ro.view(bench.VIEW, bench.VIEW, 3), rd.view(bench.VIEW, bench.VIEW, 3)
    out = []
    for name, (o, d) in (("rows", (ro.reshape(-1, 3)[400 * 800:400 * 800 + R], rd.reshape(-1, 3)[400 * 800:400 * 800 + R])),
                         ("columns", (ro.permute(1, 0, 2).reshape(-1, 3)[400 * 800:400 * 800 + R], rd.permute(1, 0, 2).reshape(-1, 3)[400 * 800:400 * 800 + R]))):
        o, d = o.contiguous(), d.contiguous()
        nears, fars = raymarching.near_far_from_aabb(o, d, m.aabb_train, m.min_near)
        t = torch.linspace(0, 1, T, device=dev)[None, :]
        z = nears[:, None] + (fars - nears)[:, None] * t
        x = ((o[:, None, :] + d[:, None, :] * z[..., None]).clamp(-m.bound, m.bound) + m.bound) / (2 * m.bound)
        out.append(f"{name}: ray-major {time_fwd(x):.3f}  64-ray blocks {time_fwd(x.view(R // 64, 64, T, 3).permute(0, 2, 1, 3)):.3f}")
    right = poses[v, :3, 0].tolist()
    print(f"pose {v} right=({right[0]:+.2f},{right[1]:+.2f},{right[2]:+.2f})  " + "   ".join(out), flush=True)
