"""Multi-object combine — the one-object-per-GPU version of COMBINED.py / editable.py.

Reference semantics (COMBINED.py:592-618, 247-251, 141-200): every object network is evaluated on the
SAME rays and the SAME T fixed-step sample positions; per sample the colour of the object with the
highest density wins (strict '>', first checkpoint wins ties); the merged (sigma, rgb) field is
composited once. `editable.py` shifts the ray origins of ONE object before its evaluation
(modify_rays_for_object) and is otherwise identical.

Sharding: objects are independent until the per-sample select, so rank r holds object
r and evaluates it locally; the select is ONE all-reduce(MAX) over an order-preserving 64-bit key
(float_bits(sigma) << 32 | 0xFFFFFFFF - rank) followed by ONE all-reduce(SUM) of rgb masked to the
winning rank (exactly one non-zero contributor per sample, so the sum is bit-exact). Ranks then
composite disjoint ray slices and all-gather the image. `mode="fast"` is the cheaper model worded in
north_star — each rank composites its own object and the per-ray (rgb, depth, weight) are summed —
which is NOT the reference's semantics (no inter-object occlusion); it is offered and labelled as such.

The pack/unpack/select/composite kernels live in libfocnerf_hip.so; `ops` can be replaced by a
CPU implementation in the world_size-2 gloo tests, which exercise only the host/collective logic.
"""
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


def modify_rays_for_object(rays_o, offset):
    """editable.py: the edited object's rays start at rays_o + (offset_x, offset_y, offset_z)."""
    return rays_o + torch.as_tensor(offset, dtype=rays_o.dtype, device=rays_o.device)


class ObjectCombiner:
    """One object per rank. All tensors live on the rank's device; collectives go over `group`
    (backend nccl == RCCL on ROCm, gloo in the CPU tests)."""

    def __init__(self, rank=None, world_size=None, group=None, ops=HipCombineOps):
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world_size is None else world_size
        self.ops = ops

    # ---- faithful per-sample select across ranks
    def select(self, dens, rgb):
        """dens [N,T] fp32 (>= 0), rgb [N,T,3] fp32 of THIS rank's object -> merged (max_dens, best_rgb), identical on all ranks."""
        dens = dens.contiguous().float()
        rgb = rgb.contiguous().float()
        keys = self.ops.pack_keys(dens, self.rank)
        if self.world > 1:
            dist.all_reduce(keys, op=dist.ReduceOp.MAX, group=self.group)
        max_dens, masked = self.ops.unpack(keys, self.rank, rgb)
        if self.world > 1:
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
        if self.world > 1:
            imgs = [torch.empty_like(img) for _ in range(self.world)]
            deps = [torch.empty_like(dep) for _ in range(self.world)]
            dist.all_gather(imgs, img, group=self.group)
            dist.all_gather(deps, dep, group=self.group)
            img, dep = torch.cat(imgs)[:N], torch.cat(deps)[:N]
        else:
            img, dep = img[:N], dep[:N]
        return img, dep

    # ---- north_star's cheaper model: per-ray sums of independently composited objects
    def render_chunk_fast(self, image, depth, weights_sum):
        """image [N,3] (premultiplied, no background), depth [N], weights_sum [N] of this rank's object.
        One all-reduce(SUM) of the packed [N,5]; the caller adds the background with the summed weight."""
        packed = torch.cat([image.float(), depth.float().unsqueeze(-1), weights_sum.float().unsqueeze(-1)], dim=-1).contiguous()
        if self.world > 1:
            dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=self.group)
        return packed[:, :3], packed[:, 3], packed[:, 4]
