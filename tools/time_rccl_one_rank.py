"""What ONE GPU can measure of the combiner's exchange: one rank on the `nccl` backend (= RCCL), the combiner issuing its
collectives anyway (`ObjectCombiner(collectives_at_world_1=True)`), on the bench's own `combined_render` workload — an 800 x 800 view x 512
samples, one FOC object, 16384-ray pieces of 134 MB. With one rank RCCL copies each piece on the device, on its own stream, while this
library's kernels evaluate the next piece: the difference to the exchange-free view is the fixed cost of the collective machinery per piece
plus what a 134 MB copy beside the field evaluation costs it (shared HBM). NOT a link measurement: no byte leaves the GPU.

    python tools/time_rccl_one_rank.py [views]          (one line per mode, then one JSON line)"""
import datetime
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29741")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist

import bench
from focnerf_amd import raymarching, synthetic
from focnerf_amd.combine import ObjectCombiner

views = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))

poses, intr = bench.make_training_rays(dev, 1, 8, seed=0)
vo, vd = synthetic.get_rays(poses[:1], intr, bench.VIEW, bench.VIEW)
vo, vd = vo[0].contiguous(), vd[0].contiguous()
n_rays, T = vo.shape[0], bench.NUM_STEPS
fn = bench.resident_object_fields(dev, 1, vo, vd, 1)[0]
probe = bench.build_foc_model(1, dev, seed=0)
nears, fars = raymarching.near_far_from_aabb(vo, vd, probe.aabb_infer, probe.min_near)
del probe


def timed(comb, overlap, chunk=16384):
    comb.render_view([fn], n_rays, nears, fars, T, bgs=(1.0, 0.0), max_ray_batch=chunk, overlap=overlap)      # untimed: allocator, RCCL channels
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(views):
        img, dep = comb.render_view([fn], n_rays, nears, fars, T, bgs=(1.0, 0.0), max_ray_batch=chunk, overlap=overlap)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / views, float(img.double().sum().item())


def eval_only(chunk=16384):
    buf = torch.empty(chunk, T, 4, dtype=torch.float32, device=dev)

    def once():
        for lo in range(0, n_rays, chunk):
            hi = min(lo + chunk, n_rays)
            fn(lo, hi, buf[: hi - lo])
    once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(views):
        once()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / views


plain, rccl = ObjectCombiner(rank=0, world_size=1), ObjectCombiner(collectives_at_world_1=True)
out = {"views": views, "rays": n_rays, "samples_per_ray": T, "piece_rays": 16384, "pieces_per_view": (n_rays + 16383) // 16384, "backend": dist.get_backend()}
for rep in range(2):                                    # interleaved: the boxes drift by a percent or two over seconds
    out[f"field_eval_only_s_{rep}"] = eval_only()
    out[f"exchange_free_s_{rep}"], cs0 = timed(plain, True)
    out[f"rccl_overlap_s_{rep}"], cs1 = timed(rccl, True)
    out[f"rccl_no_overlap_s_{rep}"], cs2 = timed(rccl, False)
    assert cs0 == cs1 == cs2, (cs0, cs1, cs2)
    print(f"rep {rep}: field evaluation alone {1e3 * out[f'field_eval_only_s_{rep}']:.2f} ms/view | exchange-free {1e3 * out[f'exchange_free_s_{rep}']:.2f} | "
          f"through RCCL, overlapped {1e3 * out[f'rccl_overlap_s_{rep}']:.2f} | through RCCL, not overlapped {1e3 * out[f'rccl_no_overlap_s_{rep}']:.2f}", flush=True)
out["image_checksum"] = cs0
print(json.dumps(out), flush=True)
dist.destroy_process_group()
