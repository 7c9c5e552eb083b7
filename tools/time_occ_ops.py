"""Library ops of the occupancy-grid training step (configs[2]) timed in isolation over several batch sizes: the intercept of
time(batch) is the fixed cost of a launch (weight staging, workspace flush, ramp and tail), the slope the per-sample cost.
Samples come from the marcher on the bench's bound-2 scene, so the access pattern is the step's own.  RAYS=1024,2048,... overrides."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from focnerf_amd import raymarching
from focnerf_amd.backend import _gridencoder, _ffmlp

dev = torch.device("cuda", 0)
m = bench.build_model(2, dev, cuda_ray=True, seed=0).train()
poses, intr = bench.make_training_rays(dev, 2, 8, seed=0)
gen = torch.Generator().manual_seed(1)
enc, sn, cn = m.encoder, m.sigma_net, m.color_net
emb = enc.embeddings.detach().half().contiguous()
ws_, wc_ = sn.weights.detach().half().contiguous(), cn.weights.detach().half().contiguous()
L = enc.offsets.shape[0] - 1
S, H = float(np.log2(enc.per_level_scale)), enc.base_resolution
REPS = int(os.environ.get("REPS", "20"))


def timed(fn):
    fn(); fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(REPS):
        fn()
    e.record(); torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / REPS          # us


rows = []
for n_rays in [int(x) for x in os.environ.get("RAYS", "1024,2048,4096,8192,16384").split(",")]:
    bench.NUM_RAYS = n_rays
    ro, rd, _ = bench.sample_batch(poses, intr, dev, gen)
    nears, fars = raymarching.near_far_from_aabb(ro[0], rd[0], m.aabb_train, m.min_near)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(ro[0], rd[0], float(m.bound), m.density_bitfield, m.cascade, 128, nears, fars, counter, -1, True, 128, False,
                                                            1 / 128, 1024)
    B = xyzs.shape[0]
    x = ((xyzs + m.bound) / (2 * m.bound)).contiguous()
    planes = torch.empty(L, B, 2, device=dev, dtype=torch.half)
    h = torch.empty(B, 16, device=dev, dtype=torch.half)
    gh = (torch.randn(B, 16, device=dev) * 1e-3).half()
    g_planes = torch.empty_like(planes)
    g_ws = torch.empty_like(ws_)
    g_emb = torch.zeros_like(emb)
    cin = torch.randn(B, 32, device=dev).half()
    cout = torch.empty(B, 16, device=dev, dtype=torch.half)
    g_cin = torch.empty_like(cin)
    g_wc = torch.empty_like(wc_)
    tick = [None]

    def fwd_counted():
        tick[0] = _gridencoder.grid_encode_forward_counted(x, emb, enc.offsets, planes, B, 3, 2, L, S, H, enc.gridtype_id, enc.align_corners, enc.interp_id)

    def fwd_plain():
        _gridencoder.grid_encode_forward(x, emb, enc.offsets, planes, B, 3, 2, L, S, H, None, enc.gridtype_id, enc.align_corners, enc.interp_id)

    def bwd_grid():
        _gridencoder.grid_encode_backward(g_planes, x, emb, enc.offsets, g_emb, B, 3, 2, L, S, H, None, None, enc.gridtype_id, enc.align_corners, enc.interp_id,
                                          grad_bl=False, precount=None)

    def fwd_bwd_grid():
        fwd_counted()
        _gridencoder.grid_encode_backward(g_planes, x, emb, enc.offsets, g_emb, B, 3, 2, L, S, H, None, None, enc.gridtype_id, enc.align_corners, enc.interp_id,
                                          grad_bl=False, precount=tick[0])

    r = {"rays": n_rays, "B": B}
    r["grid_fwd_plain"] = timed(fwd_plain)
    r["grid_fwd_counted"] = timed(fwd_counted)
    r["sigma_fwd"] = timed(lambda: _ffmlp.ffmlp_forward_planar(planes, ws_, B, 32, 16, 64, sn.num_layers, 0, 6, h))
    r["sigma_bwd"] = timed(lambda: _ffmlp.ffmlp_backward_planar(gh, planes, ws_, B, 32, 16, 64, sn.num_layers, 0, 6, True, g_planes, g_ws))
    r["grid_bwd_uncounted"] = timed(bwd_grid)
    r["grid_fwd+bwd_counted"] = timed(fwd_bwd_grid)
    r["color_fwd"] = timed(lambda: _ffmlp.ffmlp_forward(cin, wc_, B, 32, 16, 64, cn.num_layers, 0, 6, None, cout))
    r["color_bwd"] = timed(lambda: _ffmlp.ffmlp_backward(gh, cin, wc_, None, B, 32, 16, 64, cn.num_layers, 0, 6, True, None, g_cin, g_wc))
    rows.append(r)
    print(" ".join(f"{k} {v:.1f}" if isinstance(v, float) else f"{k} {v}" for k, v in r.items()), flush=True)

# least-squares line per op: us = a + b * (B / 1e6)
Bs = np.array([r["B"] for r in rows], dtype=np.float64) / 1e6
for k in rows[0]:
    if k in ("rays", "B"):
        continue
    ys = np.array([r[k] for r in rows])
    b, a = np.polyfit(Bs, ys, 1)
    print(f"{k:24s} fixed {a:7.1f} us  + {b:7.1f} us per 1e6 samples")
