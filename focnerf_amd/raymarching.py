"""raymarching ops — same public names, argument orders, defaults and return values as the
reference's raymarching/raymarching.py (:19-373), backed by libfocnerf_hip.so.

Differences that are deliberate and documented:
  * march_rays_train reserves output slots in ray order (deterministic), see include/focnerf.h;
  * no torch.cuda.empty_cache() after the force_all_rays slice (raymarching.py:231) — it only
    returns cached blocks to the driver and costs a device synchronise.
"""
import torch
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from .backend import _raymarching as _backend


def _cuda(t):
    return t if t.is_cuda else t.cuda()


# ---------------------------------------------------------------- utils

class _near_far_from_aabb(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, rays_o, rays_d, aabb, min_near=0.2):
        """rays_o/rays_d [N,3], aabb [6] -> nears [N], fars [N]  (reference raymarching.py:19-49)."""
        rays_o = _cuda(rays_o).contiguous().view(-1, 3)
        rays_d = _cuda(rays_d).contiguous().view(-1, 3)
        aabb = _cuda(aabb).contiguous()
        N = rays_o.shape[0]
        nears = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
        fars = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
        _backend.near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars)
        return nears, fars


near_far_from_aabb = _near_far_from_aabb.apply


class _sph_from_ray(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, rays_o, rays_d, radius):
        """background-sphere (theta, phi) in [-1,1]^2  (reference raymarching.py:52-80)."""
        rays_o = _cuda(rays_o).contiguous().view(-1, 3)
        rays_d = _cuda(rays_d).contiguous().view(-1, 3)
        N = rays_o.shape[0]
        coords = torch.empty(N, 2, dtype=rays_o.dtype, device=rays_o.device)
        _backend.sph_from_ray(rays_o, rays_d, radius, N, coords)
        return coords


sph_from_ray = _sph_from_ray.apply


class _morton3D(Function):
    @staticmethod
    def forward(ctx, coords):
        """int32 [N,3] in [0,128) -> int32 [N] Morton index  (reference raymarching.py:83-104)."""
        coords = _cuda(coords)
        N = coords.shape[0]
        indices = torch.empty(N, dtype=torch.int32, device=coords.device)
        _backend.morton3D(coords.int().contiguous(), N, indices)
        return indices


morton3D = _morton3D.apply


class _morton3D_invert(Function):
    @staticmethod
    def forward(ctx, indices):
        """int32 [N] -> int32 [N,3]  (reference raymarching.py:106-126)."""
        indices = _cuda(indices)
        N = indices.shape[0]
        coords = torch.empty(N, 3, dtype=torch.int32, device=indices.device)
        _backend.morton3D_invert(indices.int().contiguous(), N, coords)
        return coords


morton3D_invert = _morton3D_invert.apply


class _packbits(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, grid, thresh, bitfield=None):
        """grid [C, H^3] float -> bitfield uint8 [C*H^3/8]  (reference raymarching.py:129-155)."""
        grid = _cuda(grid).contiguous()
        C, H3 = grid.shape[0], grid.shape[1]
        N = C * H3 // 8
        if bitfield is None:
            bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
        _backend.packbits(grid, N, thresh, bitfield)
        return bitfield


packbits = _packbits.apply


# ---------------------------------------------------------------- train

class _march_rays_train(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1,
                perturb=False, align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024):
        """Occupancy-guided sample generation  (reference raymarching.py:161-235).
        Returns xyzs [M,3], dirs [M,3], deltas [M,2], rays int32 [N,3] = (ray id, offset, count)."""
        rays_o = _cuda(rays_o).contiguous().view(-1, 3)
        rays_d = _cuda(rays_d).contiguous().view(-1, 3)
        density_bitfield = _cuda(density_bitfield).contiguous()
        N = rays_o.shape[0]
        M = N * max_steps
        if not force_all_rays and mean_count > 0:
            if align > 0:
                mean_count += align - mean_count % align
            M = mean_count
        dev, dt = rays_o.device, rays_o.dtype
        xyzs = torch.zeros(M, 3, dtype=dt, device=dev)
        dirs = torch.zeros(M, 3, dtype=dt, device=dev)
        deltas = torch.zeros(M, 2, dtype=dt, device=dev)
        rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
        if step_counter is None:
            step_counter = torch.zeros(2, dtype=torch.int32, device=dev)
        noises = torch.rand(N, dtype=dt, device=dev) if perturb else torch.zeros(N, dtype=dt, device=dev)
        _backend.march_rays_train(rays_o, rays_d, density_bitfield, bound, dt_gamma, max_steps, N, C, H, M,
                                  nears.contiguous(), fars.contiguous(), xyzs, dirs, deltas, rays, step_counter, noises)
        if force_all_rays or mean_count <= 0:
            m = step_counter[0].item()  # D2H copy, as in the reference (:224)
            if align > 0:
                m += align - m % align
            xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
        return xyzs, dirs, deltas, rays


march_rays_train = _march_rays_train.apply


class _composite_rays_train(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh=1e-4):
        """Front-to-back alpha compositing per ray  (reference raymarching.py:238-269)."""
        sigmas = sigmas.contiguous()
        rgbs = rgbs.contiguous()
        deltas = deltas.contiguous()
        M, N = sigmas.shape[0], rays.shape[0]
        dev, dt = sigmas.device, sigmas.dtype
        weights_sum = torch.empty(N, dtype=dt, device=dev)
        depth = torch.empty(N, dtype=dt, device=dev)
        image = torch.empty(N, 3, dtype=dt, device=dev)
        _backend.composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image)
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, depth, image)
        ctx.dims = [M, N, T_thresh]
        return weights_sum, depth, image

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_weights_sum, grad_depth, grad_image):
        # grad_depth is ignored, as in the reference (:275)
        grad_weights_sum = grad_weights_sum.contiguous()
        grad_image = grad_image.contiguous()
        sigmas, rgbs, deltas, rays, weights_sum, depth, image = ctx.saved_tensors
        M, N, T_thresh = ctx.dims
        grad_sigmas = torch.zeros_like(sigmas)
        grad_rgbs = torch.zeros_like(rgbs)
        _backend.composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image,
                                               M, N, T_thresh, grad_sigmas, grad_rgbs)
        return grad_sigmas, grad_rgbs, None, None, None


composite_rays_train = _composite_rays_train.apply


# ---------------------------------------------------------------- inference

class _march_rays(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far,
                align=-1, perturb=False, dt_gamma=0, max_steps=1024):
        """Advance each alive ray by up to n_step occupied samples  (reference raymarching.py:297-348)."""
        rays_o = _cuda(rays_o).contiguous().view(-1, 3)
        rays_d = _cuda(rays_d).contiguous().view(-1, 3)
        M = n_alive * n_step
        if align > 0:
            M += align - (M % align)
        dev, dt = rays_o.device, rays_o.dtype
        xyzs = torch.zeros(M, 3, dtype=dt, device=dev)
        dirs = torch.zeros(M, 3, dtype=dt, device=dev)
        deltas = torch.zeros(M, 2, dtype=dt, device=dev)
        noises = torch.rand(n_alive, dtype=dt, device=dev) if perturb else torch.zeros(n_alive, dtype=dt, device=dev)
        _backend.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H,
                            density_bitfield, near, far, xyzs, dirs, deltas, noises)
        return xyzs, dirs, deltas


march_rays = _march_rays.apply


class _composite_rays(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
        """In-place incremental compositing for inference  (reference raymarching.py:351-373)."""
        _backend.composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas.contiguous(), rgbs.contiguous(),
                                deltas.contiguous(), weights_sum, depth, image)
        return tuple()


composite_rays = _composite_rays.apply


def compact_alive(rays_alive, n_alive=None):
    """Device-side, order-preserving `rays_alive[rays_alive >= 0]` (extension; the reference caller
    uses a boolean mask, legacy/nerf/renderer.py:363). Returns (compacted [n_alive], count tensor int32[1])."""
    n = rays_alive.shape[0] if n_alive is None else n_alive
    out = torch.empty_like(rays_alive)
    n_out = torch.zeros(1, dtype=torch.int32, device=rays_alive.device)
    _backend.compact_alive(rays_alive, n, out, n_out)
    return out, n_out
