"""Synthetic cameras, rays and occupancy grids for tests, smoke() and bench.py.

No dataset or detector is needed: cameras follow the reference's `rand_poses` construction
(nerf/provider.py:62-87), rays follow `get_rays` (nerf/utils.py:131-139: pixel centre + 0.5,
normalise, rotate by the pose), and the scene is an analytic soft sphere whose density fills the
cascaded 128^3 grid at cell centres exactly where `update_extra_state` would sample it
(legacy/nerf/renderer.py:469-479, no jitter). All host-side torch; works on CPU and GPU tensors.
"""
import math

import torch


def rand_poses(size, device, radius=1.0, theta_range=(math.pi / 3, 2 * math.pi / 3), phi_range=(0, 2 * math.pi), generator=None):
    def normalize(v):
        return v / (torch.norm(v, dim=-1, keepdim=True) + 1e-10)
    thetas = torch.rand(size, generator=generator) * (theta_range[1] - theta_range[0]) + theta_range[0]
    phis = torch.rand(size, generator=generator) * (phi_range[1] - phi_range[0]) + phi_range[0]
    centers = torch.stack([radius * torch.sin(thetas) * torch.sin(phis), radius * torch.cos(thetas),
                           radius * torch.sin(thetas) * torch.cos(phis)], dim=-1)
    forward_vector = -normalize(centers)
    up_vector = torch.tensor([0.0, -1.0, 0.0]).unsqueeze(0).repeat(size, 1)
    right_vector = normalize(torch.cross(forward_vector, up_vector, dim=-1))
    up_vector = normalize(torch.cross(right_vector, forward_vector, dim=-1))
    poses = torch.eye(4, dtype=torch.float).unsqueeze(0).repeat(size, 1, 1)
    poses[:, :3, :3] = torch.stack((right_vector, up_vector, forward_vector), dim=-1)
    poses[:, :3, 3] = centers
    return poses.to(device)


def intrinsics(H, W, fovy_deg=50.0):
    f = H / (2 * math.tan(math.radians(fovy_deg) / 2))
    return f, f, W / 2, H / 2


def get_rays(poses, intr, H, W, inds=None):
    """poses [B,4,4]; returns rays_o, rays_d [B, N, 3] (N = H*W or len(inds))."""
    device = poses.device
    B = poses.shape[0]
    fx, fy, cx, cy = intr
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W, device=device), torch.linspace(0, H - 1, H, device=device), indexing='ij')
    i = i.t().reshape([1, H * W]).expand([B, H * W]) + 0.5
    j = j.t().reshape([1, H * W]).expand([B, H * W]) + 0.5
    if inds is not None:
        i = torch.gather(i, -1, inds)
        j = torch.gather(j, -1, inds)
    zs = torch.ones_like(i)
    xs = (i - cx) / fx * zs
    ys = (j - cy) / fy * zs
    directions = torch.stack((xs, ys, zs), dim=-1)
    directions = directions / torch.norm(directions, dim=-1, keepdim=True)
    rays_d = directions @ poses[:, :3, :3].transpose(-1, -2)
    rays_o = poses[..., :3, 3][..., None, :].expand_as(rays_d)
    return rays_o.contiguous(), rays_d.contiguous()


def sphere_density(xyz, center, radius, sigma0=50.0, softness=0.05):
    """Smoothed indicator of a ball: sigma0 * sigmoid((radius - |x - c|) / softness)."""
    r = torch.linalg.norm(xyz - center, dim=-1)
    return sigma0 * torch.sigmoid((radius - r) / softness)


def morton3D_host(coords):
    """int64 [N,3] -> Morton index; host-side numpy/torch restatement used only to lay out synthetic grids."""
    def expand(v):
        v = (v * 0x00010001) & 0xFF0000FF
        v = (v * 0x00000101) & 0x0F00F00F
        v = (v * 0x00000011) & 0xC30C30C3
        v = (v * 0x00000005) & 0x49249249
        return v
    c = coords.to(torch.int64)
    return expand(c[..., 0]) | (expand(c[..., 1]) << 1) | (expand(c[..., 2]) << 2)


def analytic_density_grid(bound, center=(0.0, 0.0, 0.0), radius_frac=0.35, sigma0=50.0, grid_size=128, device="cpu"):
    """[cascade, H^3] density grid in Morton order, sampled at the cell centres of every cascade."""
    cascade = 1 + math.ceil(math.log2(bound))
    H = grid_size
    ar = torch.arange(H, dtype=torch.int32, device=device)
    xx, yy, zz = torch.meshgrid(ar, ar, ar, indexing='ij')
    coords = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], dim=-1)
    indices = morton3D_host(coords)
    xyzs = 2 * coords.float() / (H - 1) - 1
    c = torch.tensor(center, dtype=torch.float32, device=device)
    grid = torch.zeros(cascade, H ** 3, device=device)
    for cas in range(cascade):
        b = min(2 ** cas, bound)
        half = b / H
        cas_xyzs = xyzs * (b - half)
        grid[cas, indices] = sphere_density(cas_xyzs, c, radius_frac * bound, sigma0)
    return grid


def packbits_host(grid, thresh):
    g = (grid.reshape(-1, 8) > thresh).to(torch.uint8)
    w = (2 ** torch.arange(8, device=grid.device, dtype=torch.int32)).to(torch.uint8)
    return (g * w).sum(-1).to(torch.uint8)


def make_view_rays(H, W, bound, n_views=1, seed=0, device="cpu", radius=2.0):
    g = torch.Generator().manual_seed(seed)
    poses = rand_poses(n_views, device, radius=radius, generator=g)
    return get_rays(poses, intrinsics(H, W), H, W)
