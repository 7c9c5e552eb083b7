"""Torch restatements of the occupancy-grid maintenance (what the reference does with Python loops over the grid,
nerf/renderer.py:356-418 and legacy/nerf/renderer.py:445-536), vectorised over the whole grid in Morton order. Test infrastructure:
the HIP kernels of csrc/densitygrid.hip are checked against these statistically (different random streams) and, bit for bit, against the
CPU oracle and the reference's own fixture (tests/test_gpu_densitygrid.py)."""
import torch

from focnerf_amd import raymarching


def _cell_centres(H, device):
    """[-1, 1] coordinates of the H^3 cells in Morton order, plus their integer coordinates."""
    cells = raymarching.morton3D_invert(torch.arange(H ** 3, dtype=torch.int32, device=device))
    return 2 * cells.float() / (H - 1) - 1, cells


def mark_untrained_grid(model, poses, intrinsic, pose_chunk=8):
    fx, fy, cx, cy = intrinsic
    H, dev = model.grid_size, model.density_grid.device
    poses = torch.as_tensor(poses).to(dev).float()
    unit, _ = _cell_centres(H, dev)
    seen = torch.zeros_like(model.density_grid)
    for cas in range(model.cascade):
        bound = min(2 ** cas, model.bound)
        half = bound / H
        world = unit * (bound - half)
        for lo in range(0, poses.shape[0], pose_chunk):
            R, t = poses[lo:lo + pose_chunk, :3, :3], poses[lo:lo + pose_chunk, :3, 3]
            cam = (world.unsqueeze(0) - t.unsqueeze(1)) @ R
            inside = (cam[..., 2] > 0) & (cam[..., 0].abs() < cx / fx * cam[..., 2] + 2 * half) & (cam[..., 1].abs() < cy / fy * cam[..., 2] + 2 * half)
            seen[cas] += inside.sum(0)
    model.density_grid[seen == 0] = -1
    return seen


@torch.no_grad()
def update_extra_state(model, decay=0.95):
    H, dev = model.grid_size, model.density_grid.device
    fresh = -torch.ones_like(model.density_grid)

    def measure(cas, unit):
        bound = min(2 ** cas, model.bound)
        half = bound / H
        at = unit * (bound - half) + (torch.rand_like(unit) * 2 - 1) * half
        return (model.density(at)['sigma'].reshape(-1).detach() * model.density_scale).to(fresh.dtype)

    if model.iter_density < 16:
        unit, _ = _cell_centres(H, dev)
        for cas in range(model.cascade):
            fresh[cas] = measure(cas, unit)
    else:
        k = H ** 3 // 4
        for cas in range(model.cascade):
            anywhere = torch.randint(0, H, (k, 3), device=dev)
            occupied = torch.nonzero(model.density_grid[cas] > 0).squeeze(-1)
            occupied = occupied[torch.randint(0, occupied.shape[0], [k], device=dev)]
            cells = torch.cat([anywhere, raymarching.morton3D_invert(occupied)], dim=0)
            where = torch.cat([raymarching.morton3D(anywhere).long(), occupied], dim=0)
            fresh[cas, where] = measure(cas, 2 * cells.float() / (H - 1) - 1)
    both = (model.density_grid >= 0) & (fresh >= 0)
    model.density_grid[both] = torch.maximum(model.density_grid[both] * decay, fresh[both])
    model.mean_density = torch.mean(model.density_grid.clamp(min=0)).item()
    model.iter_density += 1
    model.density_bitfield = raymarching.packbits(model.density_grid, min(model.mean_density, model.density_thresh), model.density_bitfield)
    recent = min(16, model.local_step)
    if recent > 0:
        model.mean_count = int(model.step_counter[:recent, 0].sum().item() / recent)
    model.local_step = 0
