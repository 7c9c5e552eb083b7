"""What ONE GPU can measure of the combiner's exchange (NOTEBOOK.md, rounds 1-4 section 7): one rank on the `nccl` backend (= RCCL), the combiner issuing its
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
# DUMMY_STREAMS=k: k torch streams created (and used once) BEFORE RCCL makes its own — shifts which hardware queue RCCL's stream lands on
# (HIP deals streams onto a few hardware queues in turn; two streams on one queue run one after the other)
_dummies = [torch.cuda.Stream() for _ in range(int(os.environ.get("DUMMY_STREAMS", "0")))]
for _s in _dummies:
    with torch.cuda.stream(_s):
        torch.zeros(1, device=dev)
torch.cuda.synchronize()
# SIDE_COMPUTE=before|after: everything timed runs on a torch side stream instead of the default stream, the side stream first used before /
# after RCCL's first collective (which is when RCCL's own stream is first used)
side_mode = os.environ.get("SIDE_COMPUTE", "")
side = None
if side_mode == "before":
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        torch.zeros(1, device=dev)
    torch.cuda.synchronize()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
_t = torch.ones(4, device=dev)
dist.all_reduce(_t)                                     # RCCL's stream exists and has run from here on, as after bench.py's first barrier()
torch.cuda.synchronize()
if side_mode == "after":
    side = torch.cuda.Stream()

poses, intr = bench.make_training_rays(dev, 1, 8, seed=0)
vo, vd = synthetic.get_rays(poses[:1], intr, bench.VIEW, bench.VIEW)
vo, vd = vo[0].contiguous(), vd[0].contiguous()
# as bench.py's combined_render leg sets the view up: rays in 8 x 8 pixel tiles, no_grad, fp16 parameter copies made once per scope
from focnerf_amd.field import half_cache_scope
from focnerf_amd.rayorder import view_tiling
tile_order = view_tiling(vd)
if tile_order is not None:
    vo, vd = vo.index_select(0, tile_order), vd.index_select(0, tile_order)
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
out["ray_order"] = "8x8 pixel tiles" if tile_order is not None else "as given"
out["compute_stream"] = ("side stream, first used %s RCCL's first collective" % side_mode) if side is not None else "default stream"
out["dummy_streams_before_rccl"] = len(_dummies)
reps = int(os.environ.get("REPS", "3"))
only = os.environ.get("ONLY")                           # ONLY=overlap: nothing but the overlapped exchange (for a kernel trace, tools/rccl_overlap_from_trace.py)
if only == "overlap":
    with torch.no_grad(), half_cache_scope(), torch.cuda.stream(side if side is not None else torch.cuda.current_stream()):
        t, cs = timed(rccl, True)
    print(f"overlapped exchange only: {1e3 * t:.2f} ms/view, checksum {cs}", flush=True)
    dist.destroy_process_group()
    sys.exit(0)
for rep in range(reps):                                 # interleaved: the boxes drift by a percent or two over seconds
    with torch.no_grad(), half_cache_scope(), torch.cuda.stream(side if side is not None else torch.cuda.current_stream()):
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
