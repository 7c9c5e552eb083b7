"""Ray marching and compositing ops on the GPU (csrc/raymarching.hip).

Drop-in for the reference's raymarching/raymarching.py (:19-373): the nine public callables keep their names, positional orders,
defaults and return values. Only `composite_rays_train` has a derivative, so it is the only autograd node here; the others are plain
functions that run in fp32 outside autograd (the reference wraps each in an autograd.Function just to get that cast).

Deliberate differences:
  * `march_rays_train` hands out sample slots in ray order, deterministically (the reference reserves them with atomics,
    raymarching.cu:405-406); per-ray contents are identical;
  * no `torch.cuda.empty_cache()` after slicing the sample list (raymarching.py:231): it only returns cached blocks to the driver
    at the cost of a device synchronisation.
"""
import functools

import torch

from ._autograd import AmpOp, on_gpu
from .backend import _raymarching as _kernels


def _fp32_op(fn):
    """Run `fn` outside autograd and autocast with every floating tensor argument as fp32 on the GPU."""
    def prepare(v):
        if not torch.is_tensor(v):
            return v
        v = on_gpu(v)
        return v.float() if v.is_floating_point() else v

    @functools.wraps(fn)
    def call(*args, **kwargs):
        with torch.no_grad(), torch.autocast("cuda", enabled=False):
            return fn(*[prepare(a) for a in args], **{k: prepare(v) for k, v in kwargs.items()})
    return call


def _ray_list(t):
    return t.contiguous().view(-1, 3)


def _round_up(count, align):
    return count + (align - count % align) if align > 0 else count            # a multiple of `align` gains a whole block (raymarching.py:190,226)


def _sample_buffers(m, like):
    """xyzs [m,3], dirs [m,3], deltas [m,2], zero-initialised like the reference's three torch.zeros (raymarching.py:205-207) — as three
    views of ONE zero-filled block: one fill launch instead of three on a step that is bound by its launch count."""
    block = torch.zeros(m * 8, dtype=like.dtype, device=like.device)
    return block[: 3 * m].view(m, 3), block[3 * m: 6 * m].view(m, 3), block[6 * m:].view(m, 2)


# ---------------------------------------------------------------- geometry helpers
@_fp32_op
def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    """Slab test of every ray against the box `aabb` [6] -> (nears [N], fars [N]); misses get FLT_MAX twice (raymarching.py:19-49)."""
    o, d = _ray_list(rays_o), _ray_list(rays_d)
    n = o.shape[0]
    nears, fars = o.new_empty(n), o.new_empty(n)
    _kernels.near_far_from_aabb(o, d, aabb.contiguous(), n, min_near, nears, fars)
    return nears, fars


@_fp32_op
def sph_from_ray(rays_o, rays_d, radius):
    """(theta, phi) in [-1, 1]^2 of the point where the ray leaves the background sphere (raymarching.py:52-80)."""
    o, d = _ray_list(rays_o), _ray_list(rays_d)
    coords = o.new_empty(o.shape[0], 2)
    _kernels.sph_from_ray(o, d, radius, o.shape[0], coords)
    return coords


@_fp32_op
def morton3D(coords):
    """int [N,3] cell coordinates (< 1024) -> int32 [N] Morton codes (raymarching.py:83-104)."""
    codes = torch.empty(coords.shape[0], dtype=torch.int32, device=coords.device)
    _kernels.morton3D(coords.int().contiguous(), coords.shape[0], codes)
    return codes


@_fp32_op
def morton3D_invert(indices):
    """int [N] Morton codes -> int32 [N,3] cell coordinates (raymarching.py:106-126)."""
    cells = torch.empty(indices.shape[0], 3, dtype=torch.int32, device=indices.device)
    _kernels.morton3D_invert(indices.int().contiguous(), indices.shape[0], cells)
    return cells


@_fp32_op
def packbits(grid, thresh, bitfield=None):
    """Density grid [cascades, H^3] -> occupancy bits, eight cells per byte (raymarching.py:129-155)."""
    grid = grid.contiguous()
    n_bytes = grid.shape[0] * grid.shape[1] // 8
    if bitfield is None:
        bitfield = torch.empty(n_bytes, dtype=torch.uint8, device=grid.device)
    _kernels.packbits(grid, n_bytes, thresh, bitfield)
    return bitfield


# ---------------------------------------------------------------- training
@_fp32_op
def march_rays_train(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1, perturb=False, align=-1,
                     force_all_rays=False, dt_gamma=0, max_steps=1024):
    """Occupancy-guided sampling (raymarching.py:161-235) -> xyzs [M,3], dirs [M,3], deltas [M,2], rays int32 [N,3] = (ray, first
    sample, sample count). With a positive `mean_count` (and not `force_all_rays`) the list has exactly that many slots (rounded up to
    `align`) and rays that do not fit stay empty; otherwise it is cut to the number of samples marched, which costs one device -> host
    copy, as in the reference (:224)."""
    o, d = _ray_list(rays_o), _ray_list(rays_d)
    n = o.shape[0]
    budgeted = mean_count > 0 and not force_all_rays
    capacity = _round_up(mean_count, align) if budgeted else n * max_steps
    xyzs, dirs, deltas = _sample_buffers(capacity, o)
    rays = torch.empty(n, 3, dtype=torch.int32, device=o.device)
    if step_counter is None:
        step_counter = torch.zeros(2, dtype=torch.int32, device=o.device)
    jitter = torch.rand(n, dtype=o.dtype, device=o.device) if perturb else o.new_zeros(n)
    _kernels.march_rays_train(o, d, density_bitfield.contiguous(), bound, dt_gamma, max_steps, n, C, H, capacity, nears.contiguous(),
                              fars.contiguous(), xyzs, dirs, deltas, rays, step_counter, jitter)
    if not budgeted:
        used = _round_up(int(step_counter[0].item()), align)
        xyzs, dirs, deltas = xyzs[:used], dirs[:used], deltas[:used]
    return xyzs, dirs, deltas, rays


class CompositeTrain(AmpOp):
    """Front-to-back compositing of a ray-ordered sample list (raymarching.py:238-288): (sigmas [M], rgbs [M,3], deltas [M,2],
    rays [N,3]) -> (weights_sum [N], depth [N], image [N,3]). The derivative ignores the depth, as the reference's does (:275)."""
    cast = torch.float32

    @staticmethod
    def run(ctx, sigmas, rgbs, deltas, rays, T_thresh=1e-4):
        sigmas, rgbs, deltas = sigmas.contiguous(), rgbs.contiguous(), deltas.contiguous()
        m, n = sigmas.shape[0], rays.shape[0]
        weights_sum, depth, image = sigmas.new_empty(n), sigmas.new_empty(n), sigmas.new_empty(n, 3)
        _kernels.composite_rays_train_forward(sigmas, rgbs, deltas, rays, m, n, T_thresh, weights_sum, depth, image)
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, image)
        ctx.sizes = (m, n, T_thresh)
        ctx.set_materialize_grads(False)          # an output the loss does not read arrives as None, not as a zero-filled tensor
        return weights_sum, depth, image

    @staticmethod
    def grad(ctx, d_weights_sum, _d_depth, d_image):
        sigmas, rgbs, deltas, rays, weights_sum, image = ctx.saved_tensors
        m, n, T_thresh = ctx.sizes
        both = torch.zeros(m * 4, dtype=sigmas.dtype, device=sigmas.device)      # one fill for the two zero-initialised gradients (:283-284)
        d_sigmas, d_rgbs = both[:m], both[m:].view(m, 3)
        if d_weights_sum is None and d_image is None:
            return d_sigmas, d_rgbs, None, None, None
        d_image = torch.zeros_like(image) if d_image is None else d_image.contiguous()
        d_weights_sum = None if d_weights_sum is None else d_weights_sum.contiguous()
        _kernels.composite_rays_train_backward(d_weights_sum, d_image, sigmas, rgbs, deltas, rays, weights_sum, image, m, n, T_thresh, d_sigmas, d_rgbs)
        return d_sigmas, d_rgbs, None, None, None


composite_rays_train = CompositeTrain.apply


# ---------------------------------------------------------------- inference
@_fp32_op
def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far, align=-1, perturb=False,
               dt_gamma=0, max_steps=1024, *, noises=None):
    """Up to `n_step` further samples for each of the `n_alive` rays listed in `rays_alive` (raymarching.py:297-348); slots a ray does
    not fill keep delta 0, which `composite_rays` reads as "terminated". `noises` (keyword only, not in the reference): a [>= n_alive] fp32
    tensor to use instead of a fresh `torch.rand` / `torch.zeros` per call (the render loop hands the same zero block to every iteration)."""
    o, d = _ray_list(rays_o), _ray_list(rays_d)
    slots = _round_up(n_alive * n_step, align)
    xyzs, dirs, deltas = _sample_buffers(slots, o)
    if noises is not None:
        jitter = noises[:n_alive]
    else:
        jitter = torch.rand(n_alive, dtype=o.dtype, device=o.device) if perturb else o.new_zeros(n_alive)
    _kernels.march_rays(n_alive, n_step, rays_alive, rays_t, o, d, bound, dt_gamma, max_steps, C, H, density_bitfield, near, far, xyzs, dirs,
                        deltas, jitter)
    return xyzs, dirs, deltas


@_fp32_op
def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
    """Accumulates the new samples into weights_sum / depth / image IN PLACE, stores the new `rays_t` and marks finished rays with -1 in
    `rays_alive` (raymarching.py:351-373). Returns an empty tuple like the reference."""
    _kernels.composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas.contiguous(), rgbs.contiguous(), deltas.contiguous(),
                            weights_sum, depth, image)
    return tuple()


def compact_alive(rays_alive, n_alive=None, pad=False):
    """Order-preserving `rays_alive[rays_alive >= 0]` on the device (no reference counterpart: its caller uses a boolean mask and a host
    round trip, legacy/nerf/renderer.py:363). Returns (compacted list, int32[1] count tensor). `pad`: the entries behind the count are -1
    (dead), so that a caller who does not wait for the count can hand the whole list on: `march_rays` / `composite_rays` skip them."""
    n = rays_alive.shape[0] if n_alive is None else n_alive
    kept = torch.full_like(rays_alive, -1) if pad else torch.empty_like(rays_alive)
    # the scan kernel writes the count (an empty list gets its zero from the library): no fill here
    count = torch.empty(1, dtype=torch.int32, device=rays_alive.device)
    _kernels.compact_alive(rays_alive, n, kept, count)
    return kept, count
