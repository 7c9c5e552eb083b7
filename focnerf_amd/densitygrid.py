"""Occupancy-grid maintenance ops (csrc/densitygrid.hip): the torch code of NeRFRenderer.mark_untrained_grid and
NeRFRenderer.update_extra_state (nerf/renderer.py:356-508) as a few launches without host synchronisation.
Tensor-level wrappers; `focnerf_amd.renderer.NeRFRenderer` drives them."""
import torch

from ._lib import lib, ptr, stream_of, check, require_cuda
from .backend import _scratch


def mark_untrained_grid(poses, intrinsics, bound, cascade, H, density_grid, return_count=False):
    """poses [B,4,4] fp32 c2w; density_grid [cascade, H^3] is updated in place (-1 where no camera sees the cell)."""
    require_cuda(poses, density_grid)
    poses = poses.contiguous().float()
    assert density_grid.dtype == torch.float32 and density_grid.is_contiguous() and density_grid.shape == (cascade, H ** 3)
    fx, fy, cx, cy = [float(v) for v in intrinsics]
    count = torch.empty(cascade, H ** 3, dtype=torch.int32, device=density_grid.device) if return_count else None
    check(lib.foc_mark_untrained_grid(ptr(poses), poses.shape[0], fx, fy, cx, cy, float(bound), cascade, H, ptr(density_grid), ptr(count),
                                      stream_of(density_grid)), "mark_untrained_grid")
    return count


def grid_cells_xyz(cascade, H, bound, jitter, device):
    xyzs = torch.empty(cascade * H ** 3, 3, dtype=torch.float32, device=device)
    if jitter is not None:
        assert jitter.dtype == torch.float32 and jitter.is_contiguous() and jitter.numel() == xyzs.numel()
    check(lib.foc_grid_cells_xyz(cascade, H, float(bound), ptr(jitter), ptr(xyzs), stream_of(xyzs)), "grid_cells_xyz")
    return xyzs


def grid_update_sample(density_grid, cascade, H, bound, rand_coords, rand_pick, jitter):
    """rand_coords int32 [cascade,N,3], rand_pick fp32 [cascade,N], jitter fp32 [cascade*2N,3] -> indices int32 [cascade,2N], xyzs [cascade*2N,3]."""
    require_cuda(density_grid, rand_coords, rand_pick, jitter)
    N = rand_coords.shape[1]
    assert rand_coords.dtype == torch.int32 and rand_coords.is_contiguous() and rand_coords.shape == (cascade, N, 3)
    assert rand_pick.dtype == torch.float32 and rand_pick.is_contiguous() and rand_pick.shape == (cascade, N)
    assert jitter.dtype == torch.float32 and jitter.is_contiguous() and jitter.numel() == cascade * 2 * N * 3
    dev = density_grid.device
    indices = torch.empty(cascade, 2 * N, dtype=torch.int32, device=dev)
    xyzs = torch.empty(cascade * 2 * N, 3, dtype=torch.float32, device=dev)
    nbytes = lib.foc_grid_update_sample_workspace_bytes(cascade, H)
    ws = _scratch.get("grid_update_sample", nbytes, dev)
    check(lib.foc_grid_update_sample(ptr(density_grid), cascade, H, float(bound), N, ptr(rand_coords), ptr(rand_pick), ptr(jitter), ptr(indices), ptr(xyzs),
                                     ptr(ws), nbytes, stream_of(density_grid)), "grid_update_sample")
    return indices, xyzs


def grid_update_apply(density_grid, cascade, H, sigmas, indices, density_scale, decay, density_thresh, bitfield, mean_out=None):
    """In place on density_grid / bitfield; mean_out: fp32 device scalar receiving mean(clamp(density_grid, 0))."""
    require_cuda(density_grid, sigmas, bitfield)
    sigmas = sigmas.contiguous().float().view(-1)
    Mc = sigmas.numel() // cascade
    assert sigmas.numel() == cascade * Mc
    if indices is not None:
        assert indices.dtype == torch.int32 and indices.is_contiguous() and indices.numel() == sigmas.numel()
    assert bitfield.dtype == torch.uint8 and bitfield.numel() == cascade * H ** 3 // 8
    nbytes = lib.foc_grid_update_apply_workspace_bytes(cascade, H)
    ws = _scratch.get("grid_update_apply", nbytes, density_grid.device)
    check(lib.foc_grid_update_apply(ptr(density_grid), cascade, H, ptr(sigmas), ptr(indices), Mc, float(density_scale), float(decay), float(density_thresh),
                                    ptr(bitfield), ptr(mean_out), ptr(ws), nbytes, stream_of(density_grid)), "grid_update_apply")
