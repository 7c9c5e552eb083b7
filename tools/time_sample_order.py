"""Does the order of a render chunk's samples matter to the level-major encoder forward?  One 4096-ray x 512-step chunk of an 800x800 view,
the same 2 M points laid out ray-major ([ray][step], what k_fs_sample writes), step-major ([step][ray]), in 64-ray blocks
([block][step][64 rays]: the lanes of a wave are 64 neighbouring pixels at one depth) and shuffled."""
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
ro, rd = synthetic.get_rays(poses[:1], intr, bench.VIEW, bench.VIEW)
ro, rd = ro.view(-1, 3), rd.view(-1, 3)
T, R = 512, 4096
enc = m.encoder
emb = _half_of(enc.embeddings) if enc.embeddings.dtype != torch.half else enc.embeddings
L = enc.offsets.shape[0] - 1


def chunk_points(lo, tile=None):
    if tile is None:
        o, d = ro[lo:lo + R], rd[lo:lo + R]
    else:                                        # a 64 x 64 pixel tile, rays ordered in 8 x 8 sub-tiles
        y0, x0 = tile
        ys, xs = torch.meshgrid(torch.arange(64), torch.arange(64), indexing="ij")
        ys = ys.view(8, 8, 8, 8).permute(0, 2, 1, 3).reshape(-1)
        xs = xs.view(8, 8, 8, 8).permute(0, 2, 1, 3).reshape(-1)
        idx = ((y0 + ys) * bench.VIEW + x0 + xs).to(dev)
        o, d = ro[idx], rd[idx]
    nears, fars = raymarching.near_far_from_aabb(o.contiguous(), d.contiguous(), m.aabb_train, m.min_near)
    t = torch.linspace(0, 1, T, device=dev)[None, :]
    z = nears[:, None] + (fars - nears)[:, None] * t
    x = ((o[:, None, :] + d[:, None, :] * z[..., None]).clamp(-m.bound, m.bound) + m.bound) / (2 * m.bound)
    return x                                     # [R,T,3]


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


for name, x in (("rows 400..405 (row-major rays)", chunk_points(400 * bench.VIEW)), ("64x64 tile at the centre, 8x8 sub-tiles", chunk_points(0, (368, 368)))):
    res = {}
    res["ray-major"] = time_fwd(x)
    res["step-major"] = time_fwd(x.permute(1, 0, 2))
    res["64-ray blocks"] = time_fwd(x.view(R // 64, 64, T, 3).permute(0, 2, 1, 3))
    for rb in (16, 32, 128, 256):
        res[f"{rb}-ray blocks"] = time_fwd(x.view(R // rb, rb, T, 3).permute(0, 2, 1, 3))
    for rr, ss in ((32, 2), (16, 4), (8, 8), (4, 16)):        # a wave's lanes = rr neighbouring rays x ss consecutive depths
        res[f"{rr}x{ss} patches"] = time_fwd(x.view(R // rr, rr, T // ss, ss, 3).permute(0, 2, 1, 3, 4))
    res["shuffled"] = time_fwd(x.reshape(-1, 3)[torch.randperm(R * T, device=dev)])
    print(name, "\n   " + "\n   ".join(f"{k}: {v:.3f} ms" for k, v in res.items()), flush=True)
gen = torch.Generator().manual_seed(1)
inds = torch.randint(0, bench.VIEW * bench.VIEW, (R,), generator=gen).to(dev)
for name, ii in (("training batch, rays as drawn", inds), ("training batch, rays in pixel order", inds.sort().values)):
    o, d = ro[ii], rd[ii]
    nears, fars = raymarching.near_far_from_aabb(o.contiguous(), d.contiguous(), m.aabb_train, m.min_near)
    t = torch.linspace(0, 1, T, device=dev)[None, :]
    z = nears[:, None] + (fars - nears)[:, None] * t
    x = ((o[:, None, :] + d[:, None, :] * z[..., None]).clamp(-m.bound, m.bound) + m.bound) / (2 * m.bound)
    res = {"ray-major": time_fwd(x), "64-ray blocks": time_fwd(x.view(R // 64, 64, T, 3).permute(0, 2, 1, 3)),
           "8x8 patches": time_fwd(x.view(R // 8, 8, T // 8, 8, 3).permute(0, 2, 1, 3, 4))}
    print(name, " ".join(f"{k}: {v:.3f} ms" for k, v in res.items()), flush=True)
print("uniform random", f"{time_fwd(torch.rand(R * T, 3, device=dev)):.3f} ms")
