"""Per-level forward cost on occupancy-marched points (bound 2) vs fixed-step training points (bound 1 and bound 2), same sample count."""
import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
from focnerf_amd.backend import _gridencoder
from focnerf_amd.gridencoder import level_offsets
from focnerf_amd import raymarching
dev = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(1)
def fixed_points(bound, n_rays, T):
    m = bench.build_model(bound, dev, seed=0)
    poses, intr = bench.make_training_rays(dev, bound, 8, seed=0)
    bench.NUM_RAYS = n_rays
    ro, rd, _ = bench.sample_batch(poses, intr, dev, gen)
    ro, rd = ro.view(-1, 3), rd.view(-1, 3)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, m.aabb_train, m.min_near)
    t = torch.linspace(0, 1, T, device=dev)[None, :]
    z = nears[:, None] + (fars - nears)[:, None] * t
    return ((ro[:, None, :] + rd[:, None, :] * z[..., None]).clamp(-bound, bound).view(-1, 3) + bound) / (2 * bound), m
def occ_points(n_rays):
    m = bench.build_model(2, dev, cuda_ray=True, seed=0)
    poses, intr = bench.make_training_rays(dev, 2, 8, seed=0)
    bench.NUM_RAYS = n_rays
    ro, rd, _ = bench.sample_batch(poses, intr, dev, gen)
    nears, fars = raymarching.near_far_from_aabb(ro[0], rd[0], m.aabb_train, m.min_near)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(ro[0], rd[0], 2.0, m.density_bitfield, m.cascade, 128, nears, fars, counter, -1, True, 128, False, 1 / 128, 1024)
    return ((xyzs + 2) / 4).contiguous(), m
xo, mo = occ_points(4096)
B = xo.shape[0]
sets = [("occupancy march, bound 2", xo, mo)]
for bound in (1, 2):
    x, m = fixed_points(bound, 4096, 512)
    sets.append((f"fixed-step 512/ray, bound {bound} (first B)", x[:B].contiguous(), m))
sets.append(("fixed-step 129/ray, bound 2", fixed_points(2, 4096, 129)[0][:B].contiguous(), sets[-1][2]))
sets.append(("random", torch.rand_like(xo), mo))
print("B", B)
for kind, pts, m in sets:
    enc = m.encoder
    pls = enc.per_level_scale
    n = pts.shape[0]
    out = []
    for l in range(16):
        res = int(np.ceil(16 * pls ** l))
        off = torch.from_numpy(level_offsets(3, 1, 1.0, res, 19)).to(dev)
        table = (torch.rand(int(off[-1]), 2, device=dev) - 0.5).half()
        o = torch.empty(1, n, 2, device=dev, dtype=torch.half)
        for _ in range(2):
            _gridencoder.grid_encode_forward(pts, table, off, o, n, 3, 2, 1, 0.0, res, None, 0, False, 0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            _gridencoder.grid_encode_forward(pts, table, off, o, n, 3, 2, 1, 0.0, res, None, 0, False, 0)
        e.record(); torch.cuda.synchronize()
        out.append(1e3 * s.elapsed_time(e) / 10)
    emb = enc.embeddings.detach().half()
    L = 16
    planes = torch.empty(L, n, 2, device=dev, dtype=torch.half)
    S = float(np.log2(pls))
    def full(): _gridencoder.grid_encode_forward(pts, emb, enc.offsets, planes, n, 3, 2, L, S, 16, None, 0, False, 0)
    full(); full()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): full()
    e.record(); torch.cuda.synchronize()
    print(f"{kind:44s} n {n} full {1e3 * s.elapsed_time(e) / 10:7.1f} us | per level: " + " ".join(f"{v:.0f}" for v in out) + f" | sum {sum(out):.0f}")
