"""Per-level cost of the level-major hash-grid forward (one level at a time, so everything runs on one XCD: relative numbers)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from focnerf_amd.backend import _gridencoder
from focnerf_amd.gridencoder import level_offsets
from focnerf_amd import raymarching

dev = torch.device("cuda", 0)
m = bench.build_model(1, dev, seed=0)
poses, intr = bench.make_training_rays(dev, 1, 8, seed=0)
gen = torch.Generator().manual_seed(1)
ro, rd, _ = bench.sample_batch(poses, intr, dev, gen)
ro, rd = ro.view(-1, 3), rd.view(-1, 3)
nears, fars = raymarching.near_far_from_aabb(ro, rd, m.aabb_train, m.min_near)
t = torch.linspace(0, 1, 512, device=dev)[None, :]
z = nears[:, None] + (fars - nears)[:, None] * t
x = ((ro[:, None, :] + rd[:, None, :] * z[..., None]).clamp(-1, 1).view(-1, 3) + 1) / 2
B = x.shape[0]
pls = m.encoder.per_level_scale
from focnerf_amd import synthetic
vo, vd = synthetic.get_rays(poses[:1], intr, bench.VIEW, bench.VIEW)
vo, vd = vo.view(-1, 3)[400 * 800:400 * 800 + 4096].contiguous(), vd.view(-1, 3)[400 * 800:400 * 800 + 4096].contiguous()
vn, vf = raymarching.near_far_from_aabb(vo, vd, m.aabb_train, m.min_near)
vz = vn[:, None] + (vf - vn)[:, None] * t
vx = ((vo[:, None, :] + vd[:, None, :] * vz[..., None]).clamp(-1, 1) + 1) / 2                 # [4096,512,3]
for kind, pts in (("training rays, ray-major", x.contiguous()), ("view rows, ray-major", vx.reshape(-1, 3).contiguous()),
                  ("view rows, 64-ray blocks", vx.view(64, 64, 512, 3).permute(0, 2, 1, 3).reshape(-1, 3).contiguous()), ("random", torch.rand_like(x))):
    out = []
    for l in range(16):
        res = int(np.ceil(16 * pls ** l))
        off = torch.from_numpy(level_offsets(3, 1, 1.0, res, 19)).to(dev)
        table = (torch.rand(int(off[-1]), 2, device=dev) - 0.5).half()
        o = torch.empty(1, B, 2, device=dev, dtype=torch.half)
        for _ in range(2):
            _gridencoder.grid_encode_forward(pts, table, off, o, B, 3, 2, 1, 0.0, res, None, 0, False, 0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            _gridencoder.grid_encode_forward(pts, table, off, o, B, 3, 2, 1, 0.0, res, None, 0, False, 0)
        e.record(); torch.cuda.synchronize()
        out.append(s.elapsed_time(e) / 5)
    print(kind, "B", B, " ".join(f"{v:.3f}" for v in out))
