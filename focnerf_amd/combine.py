"""Multi-object combine — the one-object-per-GPU version of COMBINED.py / editable.py.

Reference semantics (COMBINED.py:592-618, 247-251, 141-200): every object network is evaluated on the
SAME rays and the SAME T fixed-step sample positions; per sample the colour of the object with the
highest density wins (strict '>', first checkpoint wins ties); the merged (sigma, rgb) field is
composited once per background. `editable.py` shifts the ray origins of ONE object before its evaluation
(modify_rays_for_object, :443-471) and is otherwise identical.

Sharding: objects are independent until the per-sample select, so rank r holds object r
(or a run of consecutive objects, pre-merged locally — the select is associative over the checkpoint
order) and evaluates it locally. The exchange is BY RAY (`ObjectCombiner.render_view`): every object's
per-sample field travels packed as one float4 per sample (sigma, rgb masked by the object's own weights,
`foc_fixed_field_pack`); per 4096-ray chunk ONE all-to-all hands rank q the q-th slice of the chunk's rays
from every rank — (p-1)/p x 16 B per sample leave each GPU, straight to their destination over the
point-to-point xGMI links, all seven at once — and rank q runs the select over the p objects and the
composite of ITS rays in one kernel (`foc_combine_select_composite`), so the merged field is never stored
or moved. The all-to-all of chunk c is asynchronous and overlaps the field evaluation of chunk c + 1; the
[N,4] images and depths are all-gathered once per view (12.8 MB per background). No keys, no all-reduce:
the strict-'>' rule is applied to the densities themselves, in rank order.

Kept from round 1, for callers that want the reference's merged tensors on every rank: `select` = ONE
all-reduce(MAX) over an order-preserving 64-bit key (float_bits(sigma) << 32 | 0xFFFFFFFF - rank) + ONE
all-reduce(SUM) of rgb masked to the winner (2.4x the bytes, ring-bound), and `render_chunk` on top of it.
`render_chunk_fast` is the cheaper model worded in north_star — each rank composites its own object and
the per-ray (rgb, depth, weight) are summed — which is NOT the reference's semantics (no inter-object
occlusion); it is offered and labelled as such.

The kernels live in libfocnerf_hip.so; `ops` can be replaced by a CPU implementation in the gloo tests,
which exercise only the host/collective logic.
"""
import os

import torch
import torch.distributed as dist


class HipCombineOps:
    """Device kernels (include/focnerf.h: foc_combine_*, foc_composite_fixed_steps)."""

    @staticmethod
    def select(dens, rgb, max_dens, best_rgb):
        from ._lib import lib, ptr, stream_of, check, require_cuda
        require_cuda(dens, rgb, max_dens, best_rgb)
        check(lib.foc_combine_select(ptr(dens), ptr(rgb), ptr(max_dens), ptr(best_rgb), dens.numel(), stream_of(dens)), "combine_select")

    @staticmethod
    def pack_keys(dens, rank):
        from ._lib import lib, ptr, stream_of, check, require_cuda
        require_cuda(dens)
        keys = torch.empty(dens.shape, dtype=torch.int64, device=dens.device)
        check(lib.foc_combine_pack_keys(ptr(dens), rank, ptr(keys), dens.numel(), stream_of(dens)), "combine_pack_keys")
        return keys

    @staticmethod
    def unpack(keys, rank, rgb):
        from ._lib import lib, ptr, stream_of, check, require_cuda
        require_cuda(keys, rgb)
        max_dens = torch.empty(keys.shape, dtype=torch.float32, device=keys.device)
        masked = torch.empty(rgb.shape, dtype=torch.float32, device=keys.device)
        check(lib.foc_combine_unpack(ptr(keys), rank, ptr(rgb), ptr(max_dens), ptr(masked), keys.numel(), stream_of(keys)), "combine_unpack")
        return max_dens, masked

    @staticmethod
    def composite(sigmas, rgbs, nears, fars, bg):
        from ._lib import lib, ptr, stream_of, check, require_cuda
        require_cuda(sigmas, rgbs, nears, fars)
        N, T = sigmas.shape
        image4 = torch.empty(N, 4, dtype=torch.float32, device=sigmas.device)
        depth = torch.empty(N, dtype=torch.float32, device=sigmas.device)
        check(lib.foc_composite_fixed_steps(ptr(sigmas), ptr(rgbs), ptr(nears), ptr(fars), N, T, float(bg), ptr(image4), ptr(depth),
                                            stream_of(sigmas)), "composite_fixed_steps")
        return image4, depth


    @staticmethod
    def select_composite(fields4, nears, fars, bgs, want_merged=False):
        """fields4: K tensors [N,T,4] fp32 (sigma, r, g, b) in checkpoint order -> image4 [len(bgs),N,4], depth [N] (, merged4 [N,T,4])."""
        import ctypes
        from ._lib import lib, ptr, stream_of, check, require_cuda
        require_cuda(*fields4, nears, fars)
        N, T = fields4[0].shape[0], fields4[0].shape[1]
        for f in fields4:
            if f.dtype != torch.float32 or not f.is_contiguous() or tuple(f.shape) != (N, T, 4):
                raise RuntimeError("combine select_composite: fields must be contiguous float32 [N,T,4] tensors of one shape")
        dev = fields4[0].device
        K = len(fields4)
        image4 = torch.empty(len(bgs), N, 4, dtype=torch.float32, device=dev)
        depth = torch.empty(N, dtype=torch.float32, device=dev)
        merged = torch.empty(N, T, 4, dtype=torch.float32, device=dev) if want_merged else None
        ptrs = (ctypes.c_void_p * K)(*[f.data_ptr() for f in fields4])
        cbgs = (ctypes.c_float * len(bgs))(*[float(b) for b in bgs])
        check(lib.foc_combine_select_composite(ptrs, K, ptr(nears), ptr(fars), N, T, cbgs, len(bgs), ptr(image4), ptr(depth), ptr(merged),
                                               stream_of(fields4[0])), "combine_select_composite")
        return (image4, depth, merged) if want_merged else (image4, depth)

    @staticmethod
    def select4(field4, acc4):
        from ._lib import lib, ptr, stream_of, check, require_cuda
        require_cuda(field4, acc4)
        if field4.shape != acc4.shape or field4.dtype != torch.float32 or acc4.dtype != torch.float32 or not (field4.is_contiguous() and acc4.is_contiguous()):
            raise RuntimeError("combine select4: two contiguous float32 [...,4] tensors of one shape expected")
        check(lib.foc_combine_select4(ptr(field4), ptr(acc4), field4.numel() // 4, stream_of(acc4)), "combine_select4")


    @staticmethod
    def mo_select(sigma_new, feat_new, sigma_best, feat_best):
        """In place on (sigma_best [n], feat_best [n,w]): MONeRFNetwork's running max (multiobjectnetwork.py:66-82; the new object takes ties)."""
        from ._lib import lib, ptr, stream_of, check, require_cuda
        require_cuda(sigma_new, feat_new, sigma_best, feat_best)
        ts = (sigma_new, feat_new, sigma_best, feat_best)
        if (len({t.dtype for t in ts}) != 1 or sigma_new.dtype not in (torch.float16, torch.float32) or not all(t.is_contiguous() for t in ts)
                or sigma_new.shape != sigma_best.shape or feat_new.shape != feat_best.shape or feat_new.dim() != sigma_new.dim() + 1
                or feat_new.shape[:-1] != sigma_new.shape):
            raise RuntimeError("mo_select: contiguous sigma [...], feat [..., w] pairs of one dtype (float16 or float32) expected")
        check(lib.foc_mo_select(ptr(sigma_new), ptr(feat_new), ptr(sigma_best), ptr(feat_best), sigma_new.numel(), feat_new.shape[-1],
                                sigma_new.element_size(), stream_of(sigma_best)), "mo_select")


def composite_fixed_steps(sigmas, rgbs, nears, fars, bg, ops=HipCombineOps):
    """image_depth_generation of COMBINED.py:141-200: [N,T] sigmas, [N,T,3] rgbs -> image [N,4] (rgb + sum w*sigma, clamped), depth [N]."""
    return ops.composite(sigmas.contiguous().float(), rgbs.contiguous().float(), nears.contiguous().float(), fars.contiguous().float(), bg)


def combine_serial(fields, ops=HipCombineOps):
    """Single-process combine of K objects' (densities [N,T], rgbs [N,T,3]) in checkpoint order (COMBINED.py:609-618)."""
    max_d, best = None, None
    for dens, rgb in fields:
        dens = dens.contiguous().float()
        rgb = rgb.contiguous().float()
        if max_d is None:
            max_d, best = dens.clone(), rgb.clone()
        else:
            ops.select(dens, rgb, max_d, best)
    return max_d, best


def modify_rays_for_object(rays_o, offset, rays_d=None):
    """editable.py:451-462, the edited object's branch: origins + (offset_x, offset_y, offset_z) in fp32 (in-place adds of Python floats
    on fp32 columns), directions re-normalised with F.normalize (eps 1e-12). Returns the new origins, or (origins, directions) when
    `rays_d` is given."""
    new_o = rays_o + torch.as_tensor(offset, dtype=rays_o.dtype, device=rays_o.device)
    if rays_d is None:
        return new_o
    return new_o, torch.nn.functional.normalize(rays_d, dim=-1)


_EDITABLE_OBJECT_TYPES = ('book', 'chair', 'bottle', 'cup')


def object_type_from_ckpt(ckpt):
    """editable.py:500-508: the first of ['book', 'chair', 'bottle', 'cup'] that occurs as a SUBSTRING of the checkpoint path, else None
    ('ws/cupboard/ngp.pth' is a 'cup', 'ws/notebook/...' a 'book')."""
    for t in _EDITABLE_OBJECT_TYPES:
        if t in ckpt:
            return t
    return None


class RayEditor:
    """editable.py's `modify_rays_for_object` (:443-471) with its state. The object named `edit_object` is evaluated on shifted rays
    (see above). Every OTHER object is evaluated on the rays of the FIRST view that reached this method — the reference stores them in
    `self.first_rays_o / first_rays_d` on the first call of its else-branch and returns those ever after (:465-471), for every later
    view as well. That is what the reference computes, so it is what `freeze_first_view=True` (the default) reproduces;
    `freeze_first_view=False` gives the evidently intended behaviour (unedited objects see the current view's rays)."""

    def __init__(self, edit_object, offset=(0.01, 0.01, 0.60), freeze_first_view=True):      # defaults: editable.py:76-79
        self.edit_object, self.offset, self.freeze_first_view = edit_object, tuple(float(v) for v in offset), freeze_first_view
        self.first_rays_o = self.first_rays_d = None

    def __call__(self, rays_o, rays_d, object_type):
        if object_type == self.edit_object:
            return modify_rays_for_object(rays_o, self.offset, rays_d)
        if not self.freeze_first_view:
            return rays_o, rays_d
        if self.first_rays_o is None or self.first_rays_d is None:
            self.first_rays_o, self.first_rays_d = rays_o.clone(), rays_d.clone()
        return self.first_rays_o, self.first_rays_d


def pack_field4(densities, rgbs):
    """(densities [N,T] or [N,T,1], rgbs [N,T,3]) as `run(..., return_fields=True)` returns them -> packed [N,T,4] fp32 (torch ops; the
    fused producer is fixedstep.render_field4)."""
    d = densities.reshape(rgbs.shape[0], rgbs.shape[1], 1).float()
    return torch.cat([d, rgbs.float()], dim=-1).contiguous()


def combine_packed(fields4, nears, fars, bgs=(1.0, 0.0), want_merged=False, ops=HipCombineOps):
    """K resident objects on ONE device: the object loop of COMBINED.py:598-618 and image_depth_generation for every background in one
    pass over the packed fields of a ray chunk. fields4: list of [N,T,4] in checkpoint order (more than 16: pre-merged in runs)."""
    fields4 = list(fields4)
    while len(fields4) > 16:                                   # kernel argument block holds 16 pointers
        acc = fields4[0].clone()
        for f in fields4[1:16]:
            ops.select4(f, acc)
        fields4 = [acc] + fields4[16:]
    return ops.select_composite(fields4, nears.contiguous().float(), fars.contiguous().float(), tuple(bgs), want_merged)


class ObjectCombiner:
    """One object per rank. All tensors live on the rank's device; collectives go over `group`
    (backend nccl == RCCL on ROCm, gloo in the CPU tests)."""

    def __init__(self, rank=None, world_size=None, group=None, ops=HipCombineOps, collectives_at_world_1=False):
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world_size is None else world_size
        self.ops = ops
        # A single rank needs no exchange and issues none. `collectives_at_world_1` issues every collective anyway (needs an initialised
        # process group of one rank): the rehearsal of the RCCL path — buffers, split sizes, the order of the collective's stream against
        # this library's launches — on a box with one GPU (tests/test_gpu_rccl.py). Same results either way.
        self.xch = self.world > 1 or bool(collectives_at_world_1)
        self._side = {}                # device index -> the stream render_view evaluates on while collectives are in flight
        self.bytes_sent = 0            # bytes this rank put on the wire in the last render_view

    # ---- faithful per-sample select across ranks
    def select(self, dens, rgb):
        """dens [N,T] fp32 (>= 0), rgb [N,T,3] fp32 of THIS rank's object -> merged (max_dens, best_rgb), identical on all ranks."""
        dens = dens.contiguous().float()
        rgb = rgb.contiguous().float()
        keys = self.ops.pack_keys(dens, self.rank)
        if self.xch:
            dist.all_reduce(keys, op=dist.ReduceOp.MAX, group=self.group)
        max_dens, masked = self.ops.unpack(keys, self.rank, rgb)
        if self.xch:
            dist.all_reduce(masked, op=dist.ReduceOp.SUM, group=self.group)
        return max_dens, masked

    def ray_slice(self, N):
        per = (N + self.world - 1) // self.world
        lo = min(self.rank * per, N)
        return lo, min(lo + per, N), per

    def render_chunk(self, dens, rgb, nears, fars, bg=1.0):
        """Faithful combine of one ray chunk: select across ranks, composite this rank's slice of rays,
        all-gather the [N,4] image and [N] depth. Returns (image4, depth) for the whole chunk on every rank."""
        N = dens.shape[0]
        max_dens, best = self.select(dens, rgb)
        lo, hi, per = self.ray_slice(N)
        img = torch.zeros(per, 4, dtype=torch.float32, device=dens.device)
        dep = torch.zeros(per, dtype=torch.float32, device=dens.device)
        if hi > lo:
            i4, d = self.ops.composite(max_dens[lo:hi].contiguous(), best[lo:hi].contiguous(), nears[lo:hi].contiguous().float(),
                                       fars[lo:hi].contiguous().float(), bg)
            img[: hi - lo] = i4
            dep[: hi - lo] = d
        if self.xch:
            imgs = [torch.empty_like(img) for _ in range(self.world)]
            deps = [torch.empty_like(dep) for _ in range(self.world)]
            dist.all_gather(imgs, img, group=self.group)
            dist.all_gather(deps, dep, group=self.group)
            img, dep = torch.cat(imgs)[:N], torch.cat(deps)[:N]
        else:
            img, dep = img[:N], dep[:N]
        return img, dep

    # ---- exchange by ray: all-to-all of packed fields, fused select + composite on the owner of each ray slice
    def _slice_len(self, n_rays):
        return (n_rays + self.world - 1) // self.world

    def exchange_start(self, field4, bufs=None):
        """field4 [n,T,4] fp32: this rank's (pre-merged) objects on the n rays of one chunk. Starts the all-to-all that gives rank q rays
        [q*per, (q+1)*per) of every rank's field, per = ceil(n / world) (zero rows pad a ragged chunk). Returns a handle for
        `exchange_finish`; the caller may enqueue other work (the next chunk's field evaluation) in between.
        `bufs` = (send, recv) of [world*per, T, 4] to reuse (double-buffered by render_view), else allocated here."""
        n, T = field4.shape[0], field4.shape[1]
        per = self._slice_len(n)
        if not self.xch:
            return (None, field4.contiguous(), n, per)
        if bufs is None:
            bufs = (torch.empty(self.world * per, T, 4, dtype=torch.float32, device=field4.device),
                    torch.empty(self.world * per, T, 4, dtype=torch.float32, device=field4.device))
        send, recv = bufs[0][: self.world * per], bufs[1][: self.world * per]
        if send.data_ptr() != field4.data_ptr():
            send[:n].copy_(field4)
        if self.world * per > n:
            send[n:].zero_()
        work = dist.all_to_all_single(recv, send, group=self.group, async_op=True)
        self.bytes_sent += (self.world - 1) * per * T * 16
        return (work, recv, n, per)

    def my_slice(self, values, n, per, fill):
        """This rank's rows [rank*per, (rank+1)*per) of a per-ray vector of a chunk of n rays, padded with `fill` past the chunk's end."""
        lo = min(self.rank * per, n)
        hi = min(lo + per, n)
        out = torch.full((per,), float(fill), dtype=torch.float32, device=values.device)
        out[: hi - lo] = values[lo:hi]
        return out

    def exchange_finish(self, handle, nears_mine, fars_mine, bgs=(1.0, 0.0)):
        """-> (image4 [len(bgs), per, 4], depth [per]) of THIS rank's ray slice of the chunk (rows past the chunk's end are padding).
        nears_mine / fars_mine [per]: `my_slice` of the chunk's nears / fars — of the view's own (unedited) rays, because
        image_depth_generation composites along data['rays_o'] (COMBINED.py:143-149)."""
        work, recv, n, per = handle
        if work is not None:
            work.wait()
        fields = [recv[k * per:(k + 1) * per] for k in range(self.world)] if self.world > 1 else [recv]
        return self.ops.select_composite(fields, nears_mine, fars_mine, tuple(bgs))

    def render_view(self, field_fns, n_rays, nears, fars, T, bgs=(1.0, 0.0), max_ray_batch=16384, overlap=True):
        """`_render_view` (below), on a stream of this combiner's own while collectives are in flight. Measured with one rank on RCCL
        (tools/time_rccl_one_rank.py, NOTEBOOK.md rounds 1-4 section 7): when RCCL's stream is the first stream a process uses after the default one —
        bench.py at N > 1, or a fresh COMBINED.py-style process — it shares a hardware queue with the DEFAULT stream, and an all-to-all
        "under" a field evaluation enqueued on the default stream runs strictly in turn with it (44.5 ms per view overlapped = not
        overlapped); with the evaluation on any other stream the two run side by side (43.1), whichever of the two was created first.
        To the caller nothing changes: the work is ordered behind what the caller's current stream holds, and the returned tensors
        may be used on that stream. FOC_COMBINE_SIDE_STREAM=0 evaluates on the caller's stream as before."""
        dev = nears.device
        if self.xch and overlap and dev.type == "cuda" and os.environ.get("FOC_COMBINE_SIDE_STREAM", "1") != "0":
            cur = torch.cuda.current_stream(dev)
            side = self._side.get(dev.index)
            if side is None:
                side = self._side[dev.index] = torch.cuda.Stream(dev)
            if cur != side:
                side.wait_stream(cur)
                try:
                    with torch.cuda.stream(side):
                        image4, depth = self._render_view(field_fns, n_rays, nears, fars, T, bgs, max_ray_batch, overlap)
                finally:
                    cur.wait_stream(side)              # also when a field function raised: kernels on `side` may still read the caller's buffers
                image4.record_stream(cur)              # allocated under `side`, consumed by the caller on `cur`
                depth.record_stream(cur)
                return image4, depth
        return self._render_view(field_fns, n_rays, nears, fars, T, bgs, max_ray_batch, overlap)

    def _render_view(self, field_fns, n_rays, nears, fars, T, bgs=(1.0, 0.0), max_ray_batch=16384, overlap=True):
        """One view, the loop of COMBINED.py:592-618 + compute_metrics_both_backgrounds' composites, sharded by object and exchanged by ray.
        field_fns: THIS rank's objects in checkpoint order, each `fn(lo, hi, out)` -> packed field4 [hi-lo, T, 4] fp32 of the object on
        rays lo:hi of the view (it may write into `out`, a [hi-lo, T, 4] buffer, and return it). Ranks hold consecutive runs of the
        checkpoint list (rank order = checkpoint order). nears / fars [n_rays]: of the view's own rays. `max_ray_batch` = rays per piece (field
        evaluation and exchange granularity; 16384: 134 MB per object and piece at 512 samples — 10 % faster per view than 4096-ray pieces).
        Returns (image4 [len(bgs), n_rays, 4], depth [n_rays]) on every rank."""
        dev = nears.device
        p, chunk = self.world, int(max_ray_batch)
        n_chunks = (n_rays + chunk - 1) // chunk
        if n_chunks == 0:
            return torch.zeros(len(bgs), 0, 4, device=dev), torch.zeros(0, device=dev)
        per = self._slice_len(min(chunk, n_rays))                          # slice length of a full chunk
        n_last = n_rays - (n_chunks - 1) * chunk
        per_last = self._slice_len(n_last)                                 # ... and of the (possibly shorter) last one
        # this rank's nears / fars for every chunk, once per view: row j of chunk c is ray c*chunk + rank*per_c + j
        j = torch.arange(per, device=dev)
        pc = torch.full((n_chunks, 1), per, device=dev, dtype=torch.long)
        pc[-1, 0] = per_last
        nc = torch.full((n_chunks, 1), chunk, device=dev, dtype=torch.long)
        nc[-1, 0] = n_last
        within = self.rank * pc + j[None, :]
        ok = (j[None, :] < pc) & (within < nc)
        ray = (torch.arange(n_chunks, device=dev)[:, None] * chunk + within).clamp(max=n_rays - 1)
        near_tab = torch.where(ok, nears.float()[ray], torch.ones((), device=dev)).contiguous()
        far_tab = torch.where(ok, fars.float()[ray], torch.full((), 2.0, device=dev)).contiguous()
        mine4 = torch.zeros(n_chunks, len(bgs), per, 4, dtype=torch.float32, device=dev)
        mined = torch.zeros(n_chunks, per, dtype=torch.float32, device=dev)
        bufs = [tuple(torch.empty(p * per, T, 4, dtype=torch.float32, device=dev) for _ in range(2)) for _ in range(2)] if self.xch else None
        self.bytes_sent = 0

        def finish(pending):
            handle, c = pending
            k = handle[3]
            i4, d = self.exchange_finish(handle, near_tab[c, :k], far_tab[c, :k], bgs)
            mine4[c, :, :k] = i4
            mined[c, :k] = d

        pending = None
        for c in range(n_chunks):
            lo, hi = c * chunk, min((c + 1) * chunk, n_rays)
            k = self._slice_len(hi - lo)
            out = bufs[c % 2][0][: hi - lo] if bufs is not None else None
            acc = None
            for fn in field_fns:
                f4 = fn(lo, hi, out if acc is None else None)
                if acc is None:
                    acc = f4
                else:
                    self.ops.select4(f4.contiguous(), acc)                 # this rank's later objects, same strict-'>' rule
            handle = self.exchange_start(acc, (bufs[c % 2][0][: p * k], bufs[c % 2][1][: p * k]) if bufs is not None else None)
            if pending is not None:
                finish(pending)                                            # chunk c-1: its all-to-all ran under chunk c's evaluation
            pending = (handle, c)
            if not overlap:
                finish(pending)
                pending = None
        if pending is not None:
            finish(pending)
        if not self.xch:
            return mine4.permute(1, 0, 2, 3).reshape(len(bgs), -1, 4)[:, :n_rays].contiguous(), mined.reshape(-1)[:n_rays].contiguous()
        # ONE gather per view: image rows and depths of this rank's slices in one flat buffer
        flat = torch.cat([mine4.reshape(-1), mined.reshape(-1)])
        gathered = torch.empty(p * flat.numel(), dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(gathered, flat, group=self.group)
        self.bytes_sent += (p - 1) * flat.numel() * 4
        gathered = gathered.view(p, flat.numel())
        g4 = gathered[:, : mine4.numel()].reshape(p, *mine4.shape)
        gd = gathered[:, mine4.numel():].reshape(p, *mined.shape)
        image4 = torch.empty(len(bgs), n_rays, 4, dtype=torch.float32, device=dev)
        depth = torch.empty(n_rays, dtype=torch.float32, device=dev)
        full = n_chunks - 1 if (n_last != chunk or per_last != per) else n_chunks
        if full > 0:   # [rank, chunk, bg, per, 4] -> [bg, chunk, rank*per, 4]; a chunk's rays are the first `chunk` of its p*per rows
            image4[:, : full * chunk] = g4[:, :full].permute(2, 1, 0, 3, 4).reshape(len(bgs), full, p * per, 4)[:, :, :chunk].reshape(len(bgs), -1, 4)
            depth[: full * chunk] = gd[:, :full].permute(1, 0, 2).reshape(full, p * per)[:, :chunk].reshape(-1)
        if full < n_chunks:   # the shorter last chunk was cut into slices of per_last rays
            image4[:, full * chunk:] = g4[:, full, :, :per_last].permute(1, 0, 2, 3).reshape(len(bgs), p * per_last, 4)[:, :n_last]
            depth[full * chunk:] = gd[:, full, :per_last].reshape(p * per_last)[:n_last]
        return image4, depth

    # ---- north_star's cheaper model: per-ray sums of independently composited objects
    def render_chunk_fast(self, image, depth, weights_sum):
        """image [N,3] (premultiplied, no background), depth [N], weights_sum [N] of this rank's object.
        One all-reduce(SUM) of the packed [N,5]; the caller adds the background with the summed weight."""
        packed = torch.cat([image.float(), depth.float().unsqueeze(-1), weights_sum.float().unsqueeze(-1)], dim=-1).contiguous()
        if self.xch:
            dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=self.group)
        return packed[:, :3], packed[:, 3], packed[:, 4]
