"""NeRFRenderer — host-side mirror of the reference's renderer, the CALLER of the hot path.

Follows legacy/nerf/renderer.py (the only renderer in the reference whose occupancy-grid path
runs) for `run_cuda` / `update_extra_state` (:256-376, :445-536) and
nerf/renderer.py:126-238 for the fixed-step `run` (FOC's default path: num_steps=512,
upsample_steps=0, colour queried where weights > 1e-10). It exists so the ops can be driven
end-to-end on the GPU box, where the reference tree is absent; the reference's own renderer
works unchanged on top of the same ops.

Additions over the reference caller (both optional, both default to reference behaviour):
  * `device_compaction=True` in the inference loop keeps the alive-ray list on the device
    between iterations for the compaction itself (order preserving), instead of the boolean
    mask of legacy/nerf/renderer.py:363;
  * `weight_thresh` makes the colour-query mask threshold explicit (1e-10 FOC / 1e-4 legacy);
  * density-grid maintenance (`mark_untrained_grid`, `update_extra_state`) runs on the device through
    csrc/densitygrid.hip; `FOC_FUSED_GRID_UPDATE=0` selects the torch expressions of the
    reference, which stay in this file as the parity baseline.
"""
import os
import math

import time

import torch
import torch.nn as nn

from . import raymarching


def custom_meshgrid(*args):
    return torch.meshgrid(*args, indexing='ij')


class NeRFRenderer(nn.Module):
    def __init__(self, bound=1, cuda_ray=False, density_scale=1, min_near=0.2, density_thresh=0.01, bg_radius=-1):
        super().__init__()
        self.bound = bound
        self.cascade = 1 + math.ceil(math.log2(bound))
        self.grid_size = 128
        self.density_scale = density_scale
        self.min_near = min_near
        self.density_thresh = density_thresh
        self.bg_radius = bg_radius

        aabb_train = torch.FloatTensor([-bound, -bound, -bound, bound, bound, bound])
        self.register_buffer('aabb_train', aabb_train)
        self.register_buffer('aabb_infer', aabb_train.clone())

        self.cuda_ray = cuda_ray
        if cuda_ray:
            self.register_buffer('density_grid', torch.zeros([self.cascade, self.grid_size ** 3]))
            self.register_buffer('density_bitfield', torch.zeros(self.cascade * self.grid_size ** 3 // 8, dtype=torch.uint8))
            self._mean_density = 0
            self._mean_density_dev = None            # fused update: the mean stays on the device until someone reads `mean_density`
            self.iter_density = 0
            self.register_buffer('step_counter', torch.zeros(16, 2, dtype=torch.int32))
            self.mean_count = 0
            self.local_step = 0

    @property
    def mean_density(self):
        """Python float like the reference's attribute (renderer.py:497); synchronises only if a fused update left it on the device."""
        if getattr(self, "_mean_density_dev", None) is not None:
            self._mean_density = float(self._mean_density_dev.item())
            self._mean_density_dev = None
        return self._mean_density

    @mean_density.setter
    def mean_density(self, value):
        self._mean_density = value
        self._mean_density_dev = None

    def forward(self, x, d):
        raise NotImplementedError()

    def density(self, x):
        raise NotImplementedError()

    def color(self, x, d, mask=None, **kwargs):
        raise NotImplementedError()

    def reset_extra_state(self):
        if not self.cuda_ray:
            return
        self.density_grid.zero_()
        self.mean_density = 0
        self.iter_density = 0
        self.step_counter.zero_()
        self.mean_count = 0
        self.local_step = 0

    # ------------------------------------------------------------------ fixed-step path
    def run(self, rays_o, rays_d, yolo_details=None, num_steps=512, upsample_steps=0, bg_color=None, perturb=False, weight_thresh=1e-10,
            return_fields=None, **kwargs):
        """nerf/renderer.py:126-238 (upsample_steps must be 0, FOC's setting main_nerf.py:31-32).

        `yolo_details` = (ray mask [1,N] bool, bbox, object feature) as produced by nerf/utils.py:57-154; it is handed to
        `color()` and, in training, gives the outside-mask density penalty of :165. `return_fields` (None = the reference's
        behaviour in eval mode, off in training where only image/criterion are consumed) adds `densities [N,T,1]` and
        `rgbs [N,T,3]`, the per-sample fields COMBINED.py merges."""
        assert upsample_steps == 0, "only the FOC configuration (upsample_steps=0) is implemented"
        if return_fields is None:
            return_fields = not self.training
        t_start = time.time()
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N = rays_o.shape[0]
        device = rays_o.device
        aabb = self.aabb_train if self.training else self.aabb_infer

        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, aabb, self.min_near)
        nears = nears.unsqueeze(-1)
        fars = fars.unsqueeze(-1)

        z_vals = torch.linspace(0.0, 1.0, num_steps, device=device).unsqueeze(0).expand((N, num_steps))
        z_vals = nears + (fars - nears) * z_vals
        sample_dist = (fars - nears) / num_steps
        if perturb:
            z_vals = z_vals + (torch.rand(z_vals.shape, device=device) - 0.5) * sample_dist

        xyzs = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * z_vals.unsqueeze(-1)
        xyzs = torch.min(torch.max(xyzs, aabb[:3]), aabb[3:])

        density_outputs = self.density(xyzs.reshape(-1, 3))
        criterion_outside_mask = None
        if self.training and yolo_details is not None:      # :163-165
            criterion_outside_mask = torch.norm(density_outputs['sigma'].view(N, num_steps)[~yolo_details[0].squeeze(0)] - 0)
        for k, v in density_outputs.items():
            density_outputs[k] = v.view(N, num_steps, -1)

        deltas = z_vals[..., 1:] - z_vals[..., :-1]
        deltas = torch.cat([deltas, sample_dist * torch.ones_like(deltas[..., :1])], dim=-1)
        alphas = 1 - torch.exp(-deltas * self.density_scale * density_outputs['sigma'].squeeze(-1))
        alphas_shifted = torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1)
        weights = alphas * torch.cumprod(alphas_shifted, dim=-1)[..., :-1]

        dirs = rays_d.view(-1, 1, 3).expand_as(xyzs)
        sigma_field = density_outputs['sigma']
        for k, v in density_outputs.items():
            density_outputs[k] = v.view(-1, v.shape[-1])

        mask = weights > weight_thresh
        rgbs = self.color(xyzs.reshape(-1, 3), dirs.reshape(-1, 3), mask=mask.reshape(-1), yolo_details=yolo_details, **density_outputs)
        rgbs = rgbs.view(N, -1, 3)
        t_mid = time.time()

        weights_sum = weights.sum(dim=-1)
        ori_z_vals = ((z_vals - nears) / (fars - nears)).clamp(0, 1)
        depth = torch.sum(weights * ori_z_vals, dim=-1)
        image = torch.sum(weights.unsqueeze(-1) * rgbs, dim=-2)

        if self.bg_radius > 0:
            sph = raymarching.sph_from_ray(rays_o, rays_d, self.bg_radius)
            bg_color = self.background(sph, rays_d.reshape(-1, 3))
        elif bg_color is None:
            bg_color = 1
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color

        results = {
            'depth': depth.view(*prefix),
            'image': image.view(*prefix, 3),
            'weights_sum': weights_sum,
            'criterion_outside_mask': criterion_outside_mask,
            'timing': [t_mid - t_start, time.time() - t_mid],     # host-side stage times, as the reference reports them (:190,226)
        }
        if return_fields:   # what COMBINED.py's run() hands to the combiner (:528-534)
            results['densities'] = sigma_field
            results['rgbs'] = rgbs
        return results

    # ------------------------------------------------------------------ occupancy-grid path
    def run_cuda(self, rays_o, rays_d, dt_gamma=0, bg_color=None, perturb=False, force_all_rays=False, max_steps=1024,
                 T_thresh=1e-4, device_compaction=False, **kwargs):
        """legacy/nerf/renderer.py:256-376."""
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N = rays_o.shape[0]
        device = rays_o.device

        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_train if self.training else self.aabb_infer, self.min_near)

        if self.bg_radius > 0:
            sph = raymarching.sph_from_ray(rays_o, rays_d, self.bg_radius)
            bg_color = self.background(sph, rays_d)
        elif bg_color is None:
            bg_color = 1

        results = {}
        if self.training:
            counter = self.step_counter[self.local_step % 16]
            counter.zero_()
            self.local_step += 1
            xyzs, dirs, deltas, rays = raymarching.march_rays_train(rays_o, rays_d, self.bound, self.density_bitfield, self.cascade,
                                                                    self.grid_size, nears, fars, counter, self.mean_count, perturb, 128,
                                                                    force_all_rays, dt_gamma, max_steps)
            sigmas, rgbs = self(xyzs, dirs)
            sigmas = self.density_scale * sigmas
            weights_sum, depth, image = raymarching.composite_rays_train(sigmas, rgbs, deltas, rays, T_thresh)
            image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
            depth = torch.clamp(depth - nears, min=0) / (fars - nears)
            image = image.view(*prefix, 3)
            depth = depth.view(*prefix)
            results['weights_sum'] = weights_sum
        else:
            dtype = torch.float32
            weights_sum = torch.zeros(N, dtype=dtype, device=device)
            depth = torch.zeros(N, dtype=dtype, device=device)
            image = torch.zeros(N, 3, dtype=dtype, device=device)
            n_alive = N
            rays_alive = torch.arange(n_alive, dtype=torch.int32, device=device)
            rays_t = nears.clone()
            step = 0
            while step < max_steps:
                n_alive = rays_alive.shape[0]
                if n_alive <= 0:
                    break
                n_step = max(min(N // n_alive, 8), 1)
                xyzs, dirs, deltas = raymarching.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, self.bound,
                                                            self.density_bitfield, self.cascade, self.grid_size, nears, fars, 128,
                                                            perturb if step == 0 else False, dt_gamma, max_steps)
                sigmas, rgbs = self(xyzs, dirs)
                sigmas = self.density_scale * sigmas
                raymarching.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh)
                if device_compaction:
                    out, n_out = raymarching.compact_alive(rays_alive)
                    rays_alive = out[:int(n_out.item())]
                else:
                    rays_alive = rays_alive[rays_alive >= 0]
                step += n_step
            image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
            depth = torch.clamp(depth - nears, min=0) / (fars - nears)
            image = image.view(*prefix, 3)
            depth = depth.view(*prefix)

        results['depth'] = depth
        results['image'] = image
        return results

    # ------------------------------------------------------------------ density grid maintenance
    @torch.no_grad()
    def mark_untrained_grid(self, poses, intrinsic, S=64):
        """nerf/renderer.py:356-418 / legacy :380-443: cells no training camera sees get density -1 (never marched, never updated)."""
        if not self.cuda_ray:
            return
        import numpy as np
        if isinstance(poses, np.ndarray):
            poses = torch.from_numpy(poses)
        dev = self.density_bitfield.device
        poses = poses.to(dev).float()
        if dev.type == "cuda" and os.environ.get("FOC_FUSED_GRID_UPDATE", "1") != "0":
            from . import densitygrid
            count = densitygrid.mark_untrained_grid(poses, intrinsic, self.bound, self.cascade, self.grid_size, self.density_grid, return_count=True)
        else:
            B = poses.shape[0]
            fx, fy, cx, cy = intrinsic
            X = torch.arange(self.grid_size, dtype=torch.int32, device=dev).split(S)
            count = torch.zeros_like(self.density_grid)
            for xs in X:
                for ys in X:
                    for zs in X:
                        xx, yy, zz = custom_meshgrid(xs, ys, zs)
                        coords = torch.cat([xx.reshape(-1, 1), yy.reshape(-1, 1), zz.reshape(-1, 1)], dim=-1)
                        indices = raymarching.morton3D(coords).long()
                        world_xyzs = (2 * coords.float() / (self.grid_size - 1) - 1).unsqueeze(0)
                        for cas in range(self.cascade):
                            bound = min(2 ** cas, self.bound)
                            half_grid_size = bound / self.grid_size
                            cas_world_xyzs = world_xyzs * (bound - half_grid_size)
                            head = 0
                            while head < B:
                                tail = min(head + S, B)
                                cam_xyzs = cas_world_xyzs - poses[head:tail, :3, 3].unsqueeze(1)
                                cam_xyzs = cam_xyzs @ poses[head:tail, :3, :3]
                                mask_z = cam_xyzs[:, :, 2] > 0
                                mask_x = torch.abs(cam_xyzs[:, :, 0]) < cx / fx * cam_xyzs[:, :, 2] + half_grid_size * 2
                                mask_y = torch.abs(cam_xyzs[:, :, 1]) < cy / fy * cam_xyzs[:, :, 2] + half_grid_size * 2
                                mask = (mask_z & mask_x & mask_y).sum(0).reshape(-1)
                                count[cas, indices] += mask
                                head += S
            self.density_grid[count == 0] = -1
        return count

    @torch.no_grad()
    def _update_extra_state_fused(self, decay):
        """update_extra_state through csrc/densitygrid.hip: same sampling scheme, EMA and threshold, one host read (mean_count)."""
        from . import densitygrid
        dev = self.density_bitfield.device
        C, H = self.cascade, self.grid_size
        if self.iter_density < 16:
            jitter = torch.rand(C * H ** 3, 3, device=dev)
            xyzs = densitygrid.grid_cells_xyz(C, H, self.bound, jitter, dev)
            indices = None
        else:
            N = H ** 3 // 4
            rand_coords = torch.randint(0, H, (C, N, 3), device=dev, dtype=torch.int32)
            rand_pick = torch.rand(C, N, device=dev)
            jitter = torch.rand(C * 2 * N, 3, device=dev)
            indices, xyzs = densitygrid.grid_update_sample(self.density_grid, C, H, self.bound, rand_coords, rand_pick, jitter)
        sigmas = self.density(xyzs)['sigma'].reshape(-1).detach()
        mean_dev = torch.empty(1, dtype=torch.float32, device=dev)
        densitygrid.grid_update_apply(self.density_grid, C, H, sigmas, indices, self.density_scale, decay, self.density_thresh, self.density_bitfield, mean_dev)
        self._mean_density_dev = mean_dev
        self.iter_density += 1
        total_step = min(16, self.local_step)
        if total_step > 0:
            self.mean_count = int(self.step_counter[:total_step, 0].sum().item() / total_step)
        self.local_step = 0

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128):
        """legacy/nerf/renderer.py:445-536."""
        if not self.cuda_ray:
            return
        if self.density_bitfield.is_cuda and os.environ.get("FOC_FUSED_GRID_UPDATE", "1") != "0":
            return self._update_extra_state_fused(decay)
        tmp_grid = - torch.ones_like(self.density_grid)
        dev = self.density_bitfield.device
        if self.iter_density < 16:
            X = torch.arange(self.grid_size, dtype=torch.int32, device=dev).split(S)
            Y = torch.arange(self.grid_size, dtype=torch.int32, device=dev).split(S)
            Z = torch.arange(self.grid_size, dtype=torch.int32, device=dev).split(S)
            for xs in X:
                for ys in Y:
                    for zs in Z:
                        xx, yy, zz = custom_meshgrid(xs, ys, zs)
                        coords = torch.cat([xx.reshape(-1, 1), yy.reshape(-1, 1), zz.reshape(-1, 1)], dim=-1)
                        indices = raymarching.morton3D(coords).long()
                        xyzs = 2 * coords.float() / (self.grid_size - 1) - 1
                        for cas in range(self.cascade):
                            bound = min(2 ** cas, self.bound)
                            half_grid_size = bound / self.grid_size
                            cas_xyzs = xyzs * (bound - half_grid_size)
                            cas_xyzs += (torch.rand_like(cas_xyzs) * 2 - 1) * half_grid_size
                            sigmas = self.density(cas_xyzs)['sigma'].reshape(-1).detach()
                            sigmas *= self.density_scale
                            tmp_grid[cas, indices] = sigmas.to(tmp_grid.dtype)
        else:
            N = self.grid_size ** 3 // 4
            for cas in range(self.cascade):
                coords = torch.randint(0, self.grid_size, (N, 3), device=dev)
                indices = raymarching.morton3D(coords).long()
                occ_indices = torch.nonzero(self.density_grid[cas] > 0).squeeze(-1)
                rand_mask = torch.randint(0, occ_indices.shape[0], [N], dtype=torch.long, device=dev)
                occ_indices = occ_indices[rand_mask]
                occ_coords = raymarching.morton3D_invert(occ_indices)
                indices = torch.cat([indices, occ_indices], dim=0)
                coords = torch.cat([coords, occ_coords], dim=0)
                xyzs = 2 * coords.float() / (self.grid_size - 1) - 1
                bound = min(2 ** cas, self.bound)
                half_grid_size = bound / self.grid_size
                cas_xyzs = xyzs * (bound - half_grid_size)
                cas_xyzs += (torch.rand_like(cas_xyzs) * 2 - 1) * half_grid_size
                sigmas = self.density(cas_xyzs)['sigma'].reshape(-1).detach()
                sigmas *= self.density_scale
                tmp_grid[cas, indices] = sigmas.to(tmp_grid.dtype)

        valid_mask = (self.density_grid >= 0) & (tmp_grid >= 0)
        self.density_grid[valid_mask] = torch.maximum(self.density_grid[valid_mask] * decay, tmp_grid[valid_mask])
        self.mean_density = torch.mean(self.density_grid.clamp(min=0)).item()
        self.iter_density += 1

        density_thresh = min(self.mean_density, self.density_thresh)
        self.density_bitfield = raymarching.packbits(self.density_grid, density_thresh, self.density_bitfield)

        total_step = min(16, self.local_step)
        if total_step > 0:
            self.mean_count = int(self.step_counter[:total_step, 0].sum().item() / total_step)
        self.local_step = 0

    @torch.no_grad()
    def set_density_grid(self, density_grid, density_thresh=None):
        """Install a precomputed density grid [cascade, H^3] (Morton order) and pack it — used by the
        synthetic scenes of bench.py / the tests in place of a trained grid."""
        assert self.cuda_ray
        self.density_grid.copy_(density_grid)
        self.mean_density = torch.mean(self.density_grid.clamp(min=0)).item()
        thresh = min(self.mean_density, self.density_thresh) if density_thresh is None else density_thresh
        self.density_bitfield = raymarching.packbits(self.density_grid, thresh, self.density_bitfield)

    def render(self, rays_o, rays_d, yolo_details=None, staged=False, max_ray_batch=4096, **kwargs):
        """nerf/renderer.py:511-560 (and legacy/nerf/renderer.py:539-573, which has no `yolo_details`).

        Staged rendering assembles `densities [B,N,T]` and `rgbs [B,N,T,3]` for the whole view like the reference when the
        chunks carry them (`return_fields`, default on in eval mode): 1.3 + 3.9 GB per 800x800x512 view."""
        if self.cuda_ray:
            _run = self.run_cuda                         # takes no yolo_details (renderer.py:243)
        else:
            _run = lambda o, d, **kw: self.run(o, d, yolo_details, **kw)
        B, N = rays_o.shape[:2]
        device = rays_o.device
        if staged and not self.cuda_ray:
            depth = torch.empty((B, N), device=device)
            image = torch.empty((B, N, 3), device=device)
            densities = rgbs = None
            for b in range(B):
                head = 0
                while head < N:
                    tail = min(head + max_ray_batch, N)
                    results_ = _run(rays_o[b:b + 1, head:tail], rays_d[b:b + 1, head:tail], **kwargs)
                    depth[b:b + 1, head:tail] = results_['depth']
                    image[b:b + 1, head:tail] = results_['image']
                    if 'densities' in results_:
                        if densities is None:
                            T = results_['densities'].shape[1]
                            densities = torch.empty((B, N, T), device=device)
                            rgbs = torch.empty((B, N, T, 3), device=device)
                        densities[b:b + 1, head:tail] = results_['densities'].permute(2, 0, 1)
                        rgbs[b:b + 1, head:tail] = results_['rgbs']
                    head += max_ray_batch
            results = {'depth': depth, 'image': image, 'timing': results_.get('timing')}
            if densities is not None:
                results['densities'] = densities
                results['rgbs'] = rgbs
        else:
            results = _run(rays_o, rays_d, **kwargs)
        return results
