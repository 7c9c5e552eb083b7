"""Worker of tests/test_gpu_rccl.py (NOT a test module): ONE rank on cuda:0 with the `nccl` backend (= RCCL on ROCm). Every collective the object
combiner issues for N > 1 (COMBINED.py:592-618 sharded by object, focnerf_amd/combine.py) goes through RCCL here — all_to_all_single with
the double-buffered [world*per, T, 4] pieces, all_gather_into_tensor of the view's slices, the all_reduce pair of the faithful select, the
all_gather of render_chunk, the all_reduce of render_chunk_fast — and must leave the results of the exchange-free single-rank path, bit for
bit. With one rank RCCL moves the data on the device itself: what this covers is the binding (buffers, dtypes, split sizes), that a
collective enqueued on RCCL's stream is ordered against this library's launches on both sides, and the buffer reuse of the overlapped loop;
it says nothing about xGMI.  Exit code 77: the process group could not be created on this box (nothing of this repo was reached)."""
import datetime
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ.setdefault("MASTER_PORT", "29731")   # the test passes a free one
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
try:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=60))
    probe = torch.ones(8, device=dev)
    dist.all_reduce(probe)
    torch.cuda.synchronize()
    assert float(probe.sum()) == 8.0
except Exception as e:                                       # noqa: BLE001 — whatever keeps RCCL from starting is the box's, not the product's
    print("RCCL_UNAVAILABLE", repr(e), flush=True)
    sys.exit(77)

from focnerf_amd import raymarching, synthetic
from focnerf_amd.combine import ObjectCombiner
from focnerf_amd.fixedstep import render_field4
from focnerf_amd.network import NeRFNetwork


def make_object(seed):
    torch.manual_seed(seed)
    m = NeRFNetwork(bound=1).cuda().eval()
    m.encoder.embeddings.data.uniform_(-0.5, 0.5)
    return m


K, T, chunk = 3, 64, 256
models = [make_object(40 + k) for k in range(K)]
rays_o, rays_d = synthetic.make_view_rays(40, 40, 1, 1, seed=11, device="cuda")
o, d = rays_o[0].contiguous(), rays_d[0].contiguous()       # 1600 rays: six pieces of 256 and a ragged one of 64
N = o.shape[0]
nears, fars = raymarching.near_far_from_aabb(o, d, models[0].aabb_infer, models[0].min_near)
fns = [lambda lo, hi, out, mk=mk: render_field4(mk, o[lo:hi], d[lo:hi], num_steps=T, out=out) for mk in models]

plain = ObjectCombiner(rank=0, world_size=1)
img0, dep0 = plain.render_view(fns, N, nears, fars, T, bgs=(1.0, 0.0), max_ray_batch=chunk)
assert plain.bytes_sent == 0
assert (img0[0, :, :3] != 1.0).any() and (dep0 != 0).any()            # a view with content, not a background
rccl = ObjectCombiner(collectives_at_world_1=True)                     # rank and world size from the process group
assert (rccl.rank, rccl.world, rccl.xch) == (0, 1, True)
for overlap in (True, False):
    for rep in range(3):                                               # repeated: the double buffers are reused across views
        img1, dep1 = rccl.render_view(fns, N, nears, fars, T, bgs=(1.0, 0.0), max_ray_batch=chunk, overlap=overlap)
        assert torch.equal(img1, img0), f"image differs (overlap={overlap}, view {rep})"
        assert torch.equal(dep1, dep0), f"depth differs (overlap={overlap}, view {rep})"
# one piece larger than the view, and one ray per piece's slice boundary case (chunk = 1600 exactly)
for mrb in (4096, 1600, 100):
    img1, dep1 = rccl.render_view(fns, N, nears, fars, T, bgs=(1.0, 0.0), max_ray_batch=mrb)
    ref4, refd = plain.render_view(fns, N, nears, fars, T, bgs=(1.0, 0.0), max_ray_batch=mrb)
    assert torch.equal(img1, ref4) and torch.equal(dep1, refd), f"pieces of {mrb} rays"

# the caller on a stream of its own: render_view evaluates on the combiner's stream behind it, the result is used at once on the caller's
caller = torch.cuda.Stream()
caller.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(caller):
    scale = torch.full((), 2.0, device="cuda")
    for rep in range(2):
        img1, dep1 = rccl.render_view(fns, N, nears * scale / 2, fars, T, bgs=(1.0, 0.0), max_ray_batch=chunk)
        twice = img1 * scale                                           # consumed on the caller's stream without any synchronisation in between
        assert torch.equal(twice, img0 * 2) and torch.equal(dep1, dep0), f"caller stream, view {rep}"
torch.cuda.current_stream().wait_stream(caller)
os.environ["FOC_COMBINE_SIDE_STREAM"] = "0"                           # the evaluation on the caller's stream, as before: same bits
img1, dep1 = rccl.render_view(fns, N, nears, fars, T, bgs=(1.0, 0.0), max_ray_batch=chunk)
assert torch.equal(img1, img0) and torch.equal(dep1, dep0)
del os.environ["FOC_COMBINE_SIDE_STREAM"]

# the faithful per-sample select, the chunk form and the per-ray sum model on random fields
g = torch.Generator(device="cuda").manual_seed(5)
n, t = 333, 48
dens = torch.rand(n, t, device="cuda", generator=g) * 3
dens[torch.rand(n, t, device="cuda", generator=g) < 0.3] = 0
rgb = torch.rand(n, t, 3, device="cuda", generator=g)
nr = torch.rand(n, device="cuda", generator=g) + 0.2
fr = nr + 1 + torch.rand(n, device="cuda", generator=g)
a, b = plain.select(dens, rgb), rccl.select(dens, rgb)
assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
a, b = plain.render_chunk(dens, rgb, nr, fr, 1.0), rccl.render_chunk(dens, rgb, nr, fr, 1.0)
assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
image, depth, ws = torch.rand(n, 3, device="cuda", generator=g), torch.rand(n, device="cuda", generator=g), torch.rand(n, device="cuda", generator=g)
a, b = plain.render_chunk_fast(image, depth, ws), rccl.render_chunk_fast(image, depth, ws)
assert all(torch.equal(x, y) for x, y in zip(a, b))

torch.cuda.synchronize()
print("RCCL_ONE_RANK_OK backend", dist.get_backend(), "views", 9, flush=True)
dist.destroy_process_group()
