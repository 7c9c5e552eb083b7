"""Python binding of oracle/oracle.c — the CPU restatement of the reference's hot path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by focnerf_amd/. All arrays are numpy, C-contiguous; fp16 data is
numpy float16 (bit-compatible with the uint16 half_t of oracle.c).
PARITY UNPINNED for the restated CUDA kernels (see the header of oracle.c); the fixed-step
composite and trunc_exp are pinned by tests/golden/.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")


def build(force=False):
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE], stdout=sys.stderr)       # never onto stdout: bench.py prints exactly one JSON line there
    return LIB_PATH


_lib = None
_lib_nofma = None
_use_nofma = False


def lib():
    global _lib, _lib_nofma
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB_PATH)
    if _use_nofma:
        if _lib_nofma is None:
            _lib_nofma = ctypes.CDLL(os.path.join(_HERE, "_build", "liboracle_nofma.so"))
        return _lib_nofma
    return _lib


class no_fma_policy:
    """`with oracle.no_fma_policy(): ...` — the calls inside run the ORC_NO_FMA build (no contraction at all: the other end of the unpinned
    nvcc policy, oracle.c). For tests/test_fma_policy.py, which measures how far the two ends are apart; never the checker of the HIP path."""

    def __enter__(self):
        global _use_nofma
        self.prev, _use_nofma = _use_nofma, True

    def __exit__(self, *exc):
        global _use_nofma
        _use_nofma = self.prev


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dt):
    a = np.ascontiguousarray(a, dtype=dt)
    return a


u32, u64, f32, i32 = ctypes.c_uint32, ctypes.c_uint64, ctypes.c_float, ctypes.c_int


# ---------------------------------------------------------------- raymarching
def near_far_from_aabb(rays_o, rays_d, aabb, min_near):
    rays_o, rays_d, aabb = _c(rays_o, np.float32), _c(rays_d, np.float32), _c(aabb, np.float32)
    N = rays_o.shape[0]
    nears, fars = np.empty(N, np.float32), np.empty(N, np.float32)
    lib().orc_near_far_from_aabb(_p(rays_o), _p(rays_d), _p(aabb), u32(N), f32(min_near), _p(nears), _p(fars))
    return nears, fars


def sph_from_ray(rays_o, rays_d, radius):
    rays_o, rays_d = _c(rays_o, np.float32), _c(rays_d, np.float32)
    N = rays_o.shape[0]
    coords = np.empty((N, 2), np.float32)
    lib().orc_sph_from_ray(_p(rays_o), _p(rays_d), f32(radius), u32(N), _p(coords))
    return coords


def morton3D(coords):
    coords = _c(coords, np.int32)
    N = coords.shape[0]
    out = np.empty(N, np.int32)
    lib().orc_morton3D(_p(coords), u32(N), _p(out))
    return out


def morton3D_invert(indices):
    indices = _c(indices, np.int32)
    N = indices.shape[0]
    out = np.empty((N, 3), np.int32)
    lib().orc_morton3D_invert(_p(indices), u32(N), _p(out))
    return out


def packbits(grid, thresh):
    grid = _c(grid, np.float32)
    N = grid.size // 8
    out = np.empty(N, np.uint8)
    lib().orc_packbits(_p(grid), u32(N), f32(thresh), _p(out))
    return out


# ---------------------------------------------------------------- density-grid maintenance
def mark_untrained_grid(poses, intrinsics, bound, C, H, density_grid):
    """Returns (density_grid with -1 where no camera sees the cell, count [C,H^3])."""
    poses = _c(poses, np.float32).reshape(-1, 16)
    grid = _c(density_grid, np.float32).copy()
    count = np.empty(grid.shape, np.int32)
    fx, fy, cx, cy = intrinsics
    lib().orc_mark_untrained_grid(_p(poses), u32(poses.shape[0]), f32(fx), f32(fy), f32(cx), f32(cy), f32(bound), u32(C), u32(H), _p(grid), _p(count))
    return grid, count


def grid_cells_xyz(C, H, bound, jitter=None):
    out = np.empty((C * H ** 3, 3), np.float32)
    jit = _c(jitter, np.float32) if jitter is not None else None
    lib().orc_grid_cells_xyz(u32(C), u32(H), f32(bound), _p(jit), _p(out))
    return out


def grid_update_sample(density_grid, C, H, bound, rand_coords, rand_pick, jitter):
    grid = _c(density_grid, np.float32)
    rc, rp, jit = _c(rand_coords, np.int32), _c(rand_pick, np.float32), _c(jitter, np.float32)
    N = rc.shape[1]
    idx = np.empty((C, 2 * N), np.int32)
    xyz = np.empty((C * 2 * N, 3), np.float32)
    lib().orc_grid_update_sample(_p(grid), u32(C), u32(H), f32(bound), u32(N), _p(rc), _p(rp), _p(jit), _p(idx), _p(xyz))
    return idx, xyz


def grid_update_apply(density_grid, C, H, sigmas, indices, density_scale, decay, density_thresh):
    """Returns (new density_grid, bitfield, mean_density)."""
    grid = _c(density_grid, np.float32).copy()
    sig = _c(sigmas, np.float32)
    idx = _c(indices, np.int32) if indices is not None else None
    Mc = sig.size // C
    bits = np.empty(C * H ** 3 // 8, np.uint8)
    fn = lib().orc_grid_update_apply
    fn.restype = ctypes.c_float
    mean = fn(_p(grid), u32(C), u32(H), _p(sig), _p(idx), u32(Mc), f32(density_scale), f32(decay), f32(density_thresh), _p(bits))
    return grid, bits, float(mean)


def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, C, H, M, nears, fars, noises, counter=None):
    rays_o, rays_d = _c(rays_o, np.float32), _c(rays_d, np.float32)
    grid = _c(grid, np.uint8)
    nears, fars, noises = _c(nears, np.float32), _c(fars, np.float32), _c(noises, np.float32)
    N = rays_o.shape[0]
    xyzs, dirs, deltas = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    rays = np.zeros((N, 3), np.int32)
    counter = np.zeros(2, np.int32) if counter is None else _c(counter, np.int32)
    lib().orc_march_rays_train(_p(rays_o), _p(rays_d), _p(grid), f32(bound), f32(dt_gamma), u32(max_steps), u32(N), u32(C), u32(H), u32(M),
                               _p(nears), _p(fars), _p(xyzs), _p(dirs), _p(deltas), _p(rays), _p(counter), _p(noises))
    return xyzs, dirs, deltas, rays, counter


def composite_rays_train_forward(sigmas, rgbs, deltas, rays, N_out, T_thresh):
    sigmas, rgbs, deltas, rays = _c(sigmas, np.float32), _c(rgbs, np.float32), _c(deltas, np.float32), _c(rays, np.int32)
    M, N = sigmas.shape[0], rays.shape[0]
    ws, depth, image = np.zeros(N_out, np.float32), np.zeros(N_out, np.float32), np.zeros((N_out, 3), np.float32)
    lib().orc_composite_rays_train_forward(_p(sigmas), _p(rgbs), _p(deltas), _p(rays), u32(M), u32(N), f32(T_thresh), _p(ws), _p(depth), _p(image))
    return ws, depth, image


def composite_rays_train_backward(grad_ws, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, T_thresh):
    grad_ws, grad_image = _c(grad_ws, np.float32), _c(grad_image, np.float32)
    sigmas, rgbs, deltas, rays = _c(sigmas, np.float32), _c(rgbs, np.float32), _c(deltas, np.float32), _c(rays, np.int32)
    weights_sum, image = _c(weights_sum, np.float32), _c(image, np.float32)
    M, N = sigmas.shape[0], rays.shape[0]
    gs, gc = np.zeros(M, np.float32), np.zeros((M, 3), np.float32)
    lib().orc_composite_rays_train_backward(_p(grad_ws), _p(grad_image), _p(sigmas), _p(rgbs), _p(deltas), _p(rays), _p(weights_sum), _p(image),
                                            u32(M), u32(N), f32(T_thresh), _p(gs), _p(gc))
    return gs, gc


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears, fars, noises, M=None):
    rays_alive, rays_t = _c(rays_alive, np.int32), _c(rays_t, np.float32)
    rays_o, rays_d, grid = _c(rays_o, np.float32), _c(rays_d, np.float32), _c(grid, np.uint8)
    nears, fars, noises = _c(nears, np.float32), _c(fars, np.float32), _c(noises, np.float32)
    M = n_alive * n_step if M is None else M
    xyzs, dirs, deltas = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    lib().orc_march_rays(u32(n_alive), u32(n_step), _p(rays_alive), _p(rays_t), _p(rays_o), _p(rays_d), f32(bound), f32(dt_gamma), u32(max_steps),
                         u32(C), u32(H), _p(grid), _p(nears), _p(fars), _p(xyzs), _p(dirs), _p(deltas), _p(noises))
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
    """In place on (copies of) rays_alive, rays_t, weights_sum, depth, image; returns them."""
    rays_alive, rays_t = _c(rays_alive, np.int32).copy(), _c(rays_t, np.float32).copy()
    sigmas, rgbs, deltas = _c(sigmas, np.float32), _c(rgbs, np.float32), _c(deltas, np.float32)
    weights_sum, depth, image = _c(weights_sum, np.float32).copy(), _c(depth, np.float32).copy(), _c(image, np.float32).copy()
    lib().orc_composite_rays(u32(n_alive), u32(n_step), f32(T_thresh), _p(rays_alive), _p(rays_t), _p(sigmas), _p(rgbs), _p(deltas),
                             _p(weights_sum), _p(depth), _p(image))
    return rays_alive, rays_t, weights_sum, depth, image


# ---------------------------------------------------------------- gridencoder
def _dt(a):
    return 1 if a.dtype == np.float16 else 0


def grid_level_params(level, S, H):
    scale, res = f32(0), u32(0)
    lib().orc_grid_level_params(u32(level), f32(S), u32(H), ctypes.byref(scale), ctypes.byref(res))
    return scale.value, res.value


def grid_encode_forward(inputs, embeddings, offsets, D, C, L, S, H, calc_dy_dx=False, gridtype=0, align_corners=False, interp=0, acc_mode=1):
    """Returns outputs [L,B,C] (and dy_dx [B,L,D,C])."""
    inputs = _c(inputs, np.float32)
    embeddings = np.ascontiguousarray(embeddings)
    offsets = _c(offsets, np.int32)
    B = inputs.shape[0]
    out = np.zeros((L, B, C), embeddings.dtype)
    dy = np.zeros((B, L, D, C), embeddings.dtype) if calc_dy_dx else None
    lib().orc_grid_encode_forward(_p(inputs), _p(embeddings), _p(offsets), _p(out), u32(B), u32(D), u32(C), u32(L), f32(S), u32(H), _p(dy),
                                  u32(gridtype), i32(int(align_corners)), u32(interp), i32(_dt(embeddings)), i32(acc_mode))
    return (out, dy) if calc_dy_dx else out


def grid_encode_backward(grad, inputs, offsets, n_rows, D, C, L, S, H, dy_dx=None, gridtype=0, align_corners=False, interp=0):
    """grad [L,B,C]. Returns grad_embeddings [n_rows, C] (and grad_inputs [B,D] when dy_dx is given)."""
    grad = np.ascontiguousarray(grad)
    inputs, offsets = _c(inputs, np.float32), _c(offsets, np.int32)
    B = inputs.shape[0]
    ge = np.zeros((n_rows, C), grad.dtype)
    gi = np.zeros((B, D), grad.dtype) if dy_dx is not None else None
    dyc = np.ascontiguousarray(dy_dx) if dy_dx is not None else None
    lib().orc_grid_encode_backward(_p(grad), _p(inputs), _p(offsets), _p(ge), u32(B), u32(D), u32(C), u32(L), f32(S), u32(H), _p(dyc), _p(gi),
                                   u32(gridtype), i32(int(align_corners)), u32(interp), i32(_dt(grad)))
    return (ge, gi) if dy_dx is not None else ge


def grad_total_variation(inputs, embeddings, grad, offsets, weight, D, C, L, S, H, gridtype=0, align_corners=False):
    embeddings = np.ascontiguousarray(embeddings)
    inputs = _c(inputs, embeddings.dtype)
    grad = _c(grad, embeddings.dtype).copy()
    offsets = _c(offsets, np.int32)
    B = inputs.shape[0]
    lib().orc_grad_total_variation(_p(inputs), _p(embeddings), _p(grad), _p(offsets), f32(weight), u32(B), u32(D), u32(C), u32(L), f32(S), u32(H),
                                   u32(gridtype), i32(int(align_corners)), i32(_dt(embeddings)))
    return grad


# ---------------------------------------------------------------- freqencoder
def freq_encode_forward(inputs, deg):
    inputs = _c(inputs, np.float32)
    B, D = inputs.shape
    C = D + 2 * D * deg
    out = np.empty((B, C), np.float32)
    lib().orc_freq_encode_forward(_p(inputs), u32(B), u32(D), u32(deg), u32(C), _p(out))
    return out


def freq_encode_backward(grad, outputs, D, deg):
    grad, outputs = _c(grad, np.float32), _c(outputs, np.float32)
    B, C = grad.shape
    gi = np.empty((B, D), np.float32)
    lib().orc_freq_encode_backward(_p(grad), _p(outputs), u32(B), u32(D), u32(deg), u32(C), _p(gi))
    return gi


# ---------------------------------------------------------------- ffmlp
def ffmlp_forward(inputs, weights, input_dim, hidden_dim, num_layers, activation=0, training=True, acc_mode=0):
    inputs, weights = _c(inputs, np.float16), _c(weights, np.float16)
    B = inputs.shape[0]
    fb = np.zeros((num_layers, B, hidden_dim), np.float16) if training else None
    out = np.zeros((B, 16), np.float16)
    lib().orc_ffmlp_forward(_p(inputs), _p(weights), u32(B), u32(input_dim), u32(16), u32(hidden_dim), u32(num_layers), u32(activation), _p(fb),
                            _p(out), i32(acc_mode))
    return (out, fb) if training else out


def ffmlp_backward(grad, inputs, weights, forward_buffer, input_dim, hidden_dim, num_layers, activation=0, calc_grad_inputs=True):
    grad, inputs, weights, fb = _c(grad, np.float16), _c(inputs, np.float16), _c(weights, np.float16), _c(forward_buffer, np.float16)
    B = inputs.shape[0]
    bb = np.zeros((num_layers, B, hidden_dim), np.float16)
    gi = np.zeros((B, input_dim), np.float16) if calc_grad_inputs else None
    gw = np.zeros_like(weights)
    lib().orc_ffmlp_backward(_p(grad), _p(inputs), _p(weights), _p(fb), u32(B), u32(input_dim), u32(16), u32(hidden_dim), u32(num_layers),
                             u32(activation), _p(bb), _p(gi), _p(gw))
    return gw, gi, bb


# ---------------------------------------------------------------- combine
def combine_select(dens, rgb, max_dens, best_rgb):
    dens, rgb = _c(dens, np.float32), _c(rgb, np.float32)
    max_dens, best_rgb = _c(max_dens, np.float32).copy(), _c(best_rgb, np.float32).copy()
    lib().orc_combine_select(_p(dens), _p(rgb), _p(max_dens), _p(best_rgb), u64(dens.size))
    return max_dens, best_rgb


def mo_select(sigma_new, feat_new, sigma_best, feat_best):
    """nerf/multiobjectnetwork.py:66-82 on arrays of one dtype (float16 or float32): returns the updated (sigma_best, feat_best)."""
    a, b = _c(sigma_new, np.float32), _c(sigma_best, np.float32)            # half -> float is exact: the comparison is the same
    take = np.zeros(a.size, np.uint8)
    lib().orc_mo_select_mask(_p(a), _p(b), _p(take), u64(a.size))
    take = take.astype(bool).reshape(np.shape(sigma_new))
    return np.where(take, sigma_new, sigma_best), np.where(take[..., None], feat_new, feat_best)


def composite_fixed_steps(sigmas, rgbs, nears, fars, bg, clamp01=True, want_weights=False):
    sigmas, rgbs, nears, fars = _c(sigmas, np.float32), _c(rgbs, np.float32), _c(nears, np.float32), _c(fars, np.float32)
    N, T = sigmas.shape
    image4, depth = np.empty((N, 4), np.float32), np.empty(N, np.float32)
    w = np.empty((N, T), np.float32) if want_weights else None
    lib().orc_composite_fixed_steps(_p(sigmas), _p(rgbs), _p(nears), _p(fars), u32(N), u32(T), f32(bg), i32(int(clamp01)), _p(image4), _p(depth), _p(w))
    return (image4, depth, w) if want_weights else (image4, depth)


def f2h(x):
    x = _c(x, np.float32)
    out = np.empty(x.shape, np.uint16)
    lib().orc_f2h(_p(x), _p(out), u64(x.size))
    return out.view(np.float16)


def h2f(x):
    x = np.ascontiguousarray(x).view(np.uint16)
    out = np.empty(x.shape, np.float32)
    lib().orc_h2f(_p(x), _p(out), u64(x.size))
    return out
