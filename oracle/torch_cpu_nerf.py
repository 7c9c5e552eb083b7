"""TEST / BENCHMARK INFRASTRUCTURE — never imported by focnerf_amd/.  The CPU baseline of BASELINE.json configs[0]
("pure-PyTorch path, no --tcnn / --ff"), as BASELINE.md §2 and SURVEY.md §8(d) define it: the topology of the reference's
nerf/network.py:10-210 — hash grid (L=16, C=2, base 16, log2 table 19, finest 2048 * bound) -> bias-free nn.Linear sigma net
32 -> 64 -> 16 + trunc_exp; degree-4 SH directions; colour net 31 -> 64 -> 64 -> 3 + sigmoid — in fp32 torch ops on the host cores,
driven through the fixed-step renderer math of nerf/renderer.py:126-238 (`num_steps=512, upsample_steps=0`).

Why a restatement: the reference has no CPU executable of this path. nerf/network.py imports an `encoding` module the tree does not
contain, its hash grid exists only as a CUDA extension, and NeRFRenderer.run calls the CUDA-only
`raymarching.near_far_from_aabb`. What IS the reference's own code and runs on a CPU — the network class of nerf/network.py (given
an `encoding.get_encoder`) and NeRFRenderer.run (given a near/far function) — is what this file is pinned to:
tests/golden/make_golden.py imports both from /root/reference, plugs in the encoders below and the oracle's near/far, runs them on
seeded rays and stores inputs, parameters and outputs in tests/golden/cpu_network.npz; tests/test_cpu_baseline.py requires the
restated `run` / network here to reproduce those numbers bit for bit (same torch ops in the same order), and the torch hash grid to
agree with the C oracle's (oracle.c, restating gridencoder.cu:50-245).

Pieces:
  HashGridCPU     gridencoder/grid.py:96-161 (level table, normalisation) + gridencoder.cu:87-245 (index, trilinear weights) in torch
                  index ops; autograd gives the embedding gradient (index_add of w * grad, the math of gridencoder.cu:248-340).
  SHEncoderCPU    degree-4 real SH (the published torch-ngp basis; the reference ships no SH code: PARITY UNPINNED).
  NeRFNetworkCPU  nerf/network.py:10-210 restated: same sub-module names, forward / density / color / get_params.
  run_fixed_steps nerf/renderer.py:126-238 restated (upsample_steps = 0), same order of operations.
  near_far        raymarching.cu:92-156 in torch ops (the C oracle's near_far is used to check it).
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class _trunc_exp(torch.autograd.Function):
    """activation.py:5-17: exp forward, exp(clamp(x, -15, 15)) backward."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _trunc_exp.apply


def level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners=False):
    """gridencoder/grid.py:117-131."""
    offsets, offset = [], 0
    max_params = 2 ** log2_hashmap_size
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        params_in_level = min(max_params, (resolution if align_corners else resolution + 1) ** input_dim)
        params_in_level = int(np.ceil(params_in_level / 8) * 8)
        offsets.append(offset)
        offset += params_in_level
    offsets.append(offset)
    return np.array(offsets, dtype=np.int32)


class HashGridCPU(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=None,
                 **kwargs):
        super().__init__()
        if desired_resolution is not None:                                           # grid.py:101-102
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        assert input_dim == 3
        self.input_dim, self.num_levels, self.level_dim = input_dim, num_levels, level_dim
        self.per_level_scale, self.base_resolution, self.log2_hashmap_size = per_level_scale, base_resolution, log2_hashmap_size
        self.output_dim = num_levels * level_dim
        offs = level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size)
        self.register_buffer('offsets', torch.from_numpy(offs))
        self.n_params = int(offs[-1]) * level_dim
        self.embeddings = nn.Parameter(torch.empty(int(offs[-1]), level_dim))
        self.embeddings.data.uniform_(-1e-4, 1e-4)                                   # grid.py:138-140
        self._host_offsets = [int(v) for v in offs]

    def forward(self, inputs, bound=1):
        x = (inputs + bound) / (2 * bound)                                           # grid.py:149
        prefix = list(x.shape[:-1])
        x = x.reshape(-1, 3)
        S = np.float32(np.log2(self.per_level_scale))                                # the float the kernel receives (grid.py:153)
        inside = ((x >= 0) & (x <= 1)).all(-1, keepdim=True)                         # gridencoder.cu:110-135: a point outside [0,1]^3 encodes to zeros
        outs = []
        for lvl in range(self.num_levels):
            size = self._host_offsets[lvl + 1] - self._host_offsets[lvl]
            scale = float(np.float32(np.exp2(np.float32(lvl) * S)) * np.float32(self.base_resolution) - np.float32(1.0))
            res = int(math.ceil(scale)) + 1
            pos = x * scale + 0.5
            pg = torch.floor(pos)
            frac = pos - pg
            pg = pg.long()
            stride1, stride2 = res + 1, (res + 1) * (res + 1)
            dense = stride2 * (res + 1) <= size                                      # the stride loop of :66-74 runs to the end without overflowing
            acc = 0
            for corner in range(8):
                w = 1.0
                c = []
                for dim in range(3):
                    if corner & (1 << dim):
                        w = w * frac[:, dim]
                        c.append(pg[:, dim] + 1)
                    else:
                        w = w * (1 - frac[:, dim])
                        c.append(pg[:, dim])
                if dense:
                    idx = c[0] + c[1] * stride1 + c[2] * stride2
                else:                                                                 # fast_hash, uint32 wrap-around (:50-64)
                    idx = (c[0] & 0xFFFFFFFF) ^ ((c[1] * 2654435761) & 0xFFFFFFFF) ^ ((c[2] * 805459861) & 0xFFFFFFFF)
                idx = idx % size + self._host_offsets[lvl]
                acc = acc + w.unsqueeze(-1) * self.embeddings[idx]
            outs.append(acc)
        out = torch.cat(outs, dim=-1) * inside.to(x.dtype)
        return out.view(prefix + [self.output_dim])


def sh_encode_deg4(d):
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    xy, xz, yz = x * y, x * z, y * z
    x2, y2, z2 = x * x, y * y, z * z
    return torch.stack([
        torch.full_like(x, 0.28209479177387814), -0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x,
        1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.94617469575755997 * z2 - 0.31539156525251999, -1.0925484305920792 * xz,
        0.54627421529603959 * x2 - 0.54627421529603959 * y2, 0.59004358992664352 * y * (-3.0 * x2 + y2), 2.8906114426405538 * xy * z,
        0.45704579946446572 * y * (1.0 - 5.0 * z2), 0.3731763325901154 * z * (5.0 * z2 - 3.0), 0.45704579946446572 * x * (1.0 - 5.0 * z2),
        1.4453057213202769 * z * (x2 - y2), 0.59004358992664352 * x * (-x2 + 3.0 * y2)], dim=-1)


class SHEncoderCPU(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        assert input_dim == 3 and degree == 4
        self.output_dim = 16

    def forward(self, d, **kwargs):
        return sh_encode_deg4(d.float())


def get_encoder(encoding, input_dim=3, degree=4, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048, **kwargs):
    """The `encoding.get_encoder` the reference's nerf/network.py:5 imports, serving the two encodings of the baseline configuration."""
    if encoding == 'hashgrid':
        enc = HashGridCPU(input_dim, num_levels, level_dim, 2, base_resolution, log2_hashmap_size, desired_resolution)
    elif encoding == 'sphere_harmonics':
        enc = SHEncoderCPU(input_dim, degree)
    else:
        raise NotImplementedError(encoding)
    return enc, enc.output_dim


def near_far(rays_o, rays_d, aabb, min_near):
    """raymarching.cu:92-156 in torch ops: slab test, min_near clamp, FLT_MAX for both when the ray misses."""
    rd = 1.0 / rays_d
    t0, t1 = (aabb[:3] - rays_o) * rd, (aabb[3:] - rays_o) * rd
    lo, hi = torch.minimum(t0, t1), torch.maximum(t0, t1)
    near, far = lo.max(dim=-1).values, hi.min(dim=-1).values
    miss = far < near
    near = torch.where(near < min_near, torch.full_like(near, min_near), near)
    big = torch.full_like(near, torch.finfo(torch.float32).max)
    return torch.where(miss, big, near), torch.where(miss, big, far)


class NeRFNetworkCPU(nn.Module):
    """nerf/network.py:10-210 (bg_radius <= 0) + the buffers of nerf/renderer.py the fixed-step path reads."""

    def __init__(self, num_layers=2, hidden_dim=64, geo_feat_dim=15, num_layers_color=3, hidden_dim_color=64, bound=1, min_near=0.2, density_scale=1,
                 encoder_kwargs=None):
        super().__init__()
        self.bound, self.min_near, self.density_scale = bound, min_near, density_scale
        box = torch.tensor([-bound] * 3 + [bound] * 3, dtype=torch.float32)
        self.register_buffer('aabb_train', box)
        self.register_buffer('aabb_infer', box.clone())
        self.num_layers, self.hidden_dim, self.geo_feat_dim = num_layers, hidden_dim, geo_feat_dim
        self.encoder, self.in_dim = get_encoder('hashgrid', desired_resolution=2048 * bound, **(encoder_kwargs or {}))
        self.sigma_net = nn.ModuleList([nn.Linear(self.in_dim if l == 0 else hidden_dim, 1 + geo_feat_dim if l == num_layers - 1 else hidden_dim, bias=False)
                                        for l in range(num_layers)])
        self.num_layers_color, self.hidden_dim_color = num_layers_color, hidden_dim_color
        self.encoder_dir, self.in_dim_dir = get_encoder('sphere_harmonics')
        self.color_net = nn.ModuleList([nn.Linear(self.in_dim_dir + geo_feat_dim if l == 0 else hidden_dim_color,
                                                  3 if l == num_layers_color - 1 else hidden_dim_color, bias=False) for l in range(num_layers_color)])

    def density(self, x):
        h = self.encoder(x, bound=self.bound)
        for l in range(self.num_layers):
            h = self.sigma_net[l](h)
            if l != self.num_layers - 1:
                h = F.relu(h, inplace=True)
        return {'sigma': trunc_exp(h[..., 0]), 'geo_feat': h[..., 1:]}

    def color(self, x, d, mask=None, geo_feat=None, **kwargs):
        if mask is not None:
            rgbs = torch.zeros(mask.shape[0], 3, dtype=x.dtype, device=x.device)
            if not mask.any():
                return rgbs
            x, d, geo_feat = x[mask], d[mask], geo_feat[mask]
        h = torch.cat([self.encoder_dir(d), geo_feat], dim=-1)
        for l in range(self.num_layers_color):
            h = self.color_net[l](h)
            if l != self.num_layers_color - 1:
                h = F.relu(h, inplace=True)
        h = torch.sigmoid(h)
        if mask is not None:
            rgbs[mask] = h.to(rgbs.dtype)
            return rgbs
        return h

    def get_params(self, lr):
        return [{'params': m.parameters(), 'lr': lr} for m in (self.encoder, self.sigma_net, self.encoder_dir, self.color_net)]


def run_fixed_steps(model, rays_o, rays_d, num_steps=512, bg_color=None, perturb=False, near_far_fn=near_far):
    """nerf/renderer.py:126-238 with upsample_steps = 0, in its order of operations (the yolo-mask criterion of :163-165 needs the
    trainer's masks and is not part of the baseline)."""
    prefix = rays_o.shape[:-1]
    rays_o = rays_o.contiguous().view(-1, 3)
    rays_d = rays_d.contiguous().view(-1, 3)
    N = rays_o.shape[0]
    aabb = model.aabb_train if model.training else model.aabb_infer
    nears, fars = near_far_fn(rays_o, rays_d, aabb, model.min_near)
    nears, fars = nears.unsqueeze(-1), fars.unsqueeze(-1)
    z_vals = torch.linspace(0.0, 1.0, num_steps).unsqueeze(0).expand((N, num_steps))
    z_vals = nears + (fars - nears) * z_vals
    sample_dist = (fars - nears) / num_steps
    if perturb:
        z_vals = z_vals + (torch.rand(z_vals.shape) - 0.5) * sample_dist
    xyzs = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * z_vals.unsqueeze(-1)
    xyzs = torch.min(torch.max(xyzs, aabb[:3]), aabb[3:])
    density_outputs = model.density(xyzs.reshape(-1, 3))
    for k, v in density_outputs.items():
        density_outputs[k] = v.view(N, num_steps, -1)
    densities = density_outputs['sigma']
    deltas = z_vals[..., 1:] - z_vals[..., :-1]
    deltas = torch.cat([deltas, sample_dist * torch.ones_like(deltas[..., :1])], dim=-1)
    alphas = 1 - torch.exp(-deltas * model.density_scale * density_outputs['sigma'].squeeze(-1))
    alphas_shifted = torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1)
    weights = alphas * torch.cumprod(alphas_shifted, dim=-1)[..., :-1]
    dirs = rays_d.view(-1, 1, 3).expand_as(xyzs)
    for k, v in density_outputs.items():
        density_outputs[k] = v.view(-1, v.shape[-1])
    mask = weights > 1e-10
    rgbs = model.color(xyzs.reshape(-1, 3), dirs.reshape(-1, 3), mask=mask.reshape(-1), **density_outputs)
    rgbs = rgbs.view(N, -1, 3)
    weights_sum = weights.sum(dim=-1)
    ori_z_vals = ((z_vals - nears) / (fars - nears)).clamp(0, 1)
    depth = torch.sum(weights * ori_z_vals, dim=-1)
    image = torch.sum(weights.unsqueeze(-1) * rgbs, dim=-2)
    if bg_color is None:
        bg_color = 1
    image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
    return {'depth': depth.view(*prefix), 'image': image.view(*prefix, 3), 'weights_sum': weights_sum, 'densities': densities, 'rgbs': rgbs}


# ---------------------------------------------------------------------------------------------------------------- timing (bench.py)
def _view_rays(side, bound, seed=0, radius=2.0):
    """Pinhole camera of SURVEY.md §8(d): fovy 50 deg, pixel centres, one pose on the sphere of `radius` looking at the origin."""
    g = torch.Generator().manual_seed(seed)
    theta = torch.rand(1, generator=g) * (math.pi / 3) + math.pi / 3
    phi = torch.rand(1, generator=g) * 2 * math.pi
    centre = torch.stack([radius * torch.sin(theta) * torch.sin(phi), radius * torch.cos(theta), radius * torch.sin(theta) * torch.cos(phi)], -1)[0]
    fwd = -centre / centre.norm()
    right = torch.linalg.cross(fwd, torch.tensor([0.0, -1.0, 0.0]))
    right = right / right.norm()
    up = torch.linalg.cross(right, fwd)
    f = side / (2 * math.tan(math.radians(25.0)))
    j, i = torch.meshgrid(torch.arange(side, dtype=torch.float32), torch.arange(side, dtype=torch.float32), indexing='ij')
    dirs = (i.reshape(-1, 1) + 0.5 - side / 2) / f * right + (j.reshape(-1, 1) + 0.5 - side / 2) / f * up + fwd
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    return centre.expand_as(dirs).contiguous(), dirs.contiguous()


def time_baseline(render_budget_s=None, train_steps=5, train_rays=4096, side=400, num_steps=512, chunk=4096, bound=2, threads=None, budget_s=210.0):
    """configs[0] on the host cores by the protocol of BASELINE.md section 2: (a) render — ONE WHOLE side x side view (160 000 rays x 512 =
    81.9 M samples) in 4096-ray chunks through NeRFNetworkCPU + run_fixed_steps under no_grad; (b) train — the MEDIAN OF `train_steps` = 5
    FULL steps of `train_rays` = 4096 rays x 512 samples (forward + MSE + backward + Adam(betas 0.9/0.99, eps 1e-15)) after one small
    warm-up step. About 45 s + 5 x 15 s on 16 cores of an EPYC 9575F. `budget_s` is a safety net for a slower host, not the plan: when
    the view or the steps would overrun it, what has been timed by then is reported with `extrapolated: true` and the counts that were
    reached (`render_budget_s`, if given, bounds the view the same way — the smoke test of the 8-core build container uses it)."""
    import os
    import time
    t_begin = time.perf_counter()
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # the box may show every host core in the affinity mask while a cgroup quota grants far fewer (16 on the one-GPU boxes): more
    # OpenMP threads than granted cores spin against each other and a 10 s sample turns into minutes
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(float(q) / float(per)))
    except (OSError, ValueError):
        pass
    threads = threads or max(1, min(avail, quota or avail, 16))
    old_threads = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        torch.manual_seed(0)
        model = NeRFNetworkCPU(bound=bound)
        model.encoder.embeddings.data.uniform_(-0.5, 0.5)
        rays_o, rays_d = _view_rays(side, bound)
        n_view = rays_o.shape[0]
        model.eval()
        with torch.no_grad():
            run_fixed_steps(model, rays_o[:chunk], rays_d[:chunk], num_steps)        # warm-up
            done, t0 = 0, time.perf_counter()
            lo = 0
            view_limit = render_budget_s if render_budget_s is not None else 0.45 * budget_s
            while lo < n_view and time.perf_counter() - t0 < view_limit:
                hi = min(lo + chunk, n_view)
                run_fixed_steps(model, rays_o[lo:hi], rays_d[lo:hi], num_steps)
                done += hi - lo
                lo = hi
            el_r = time.perf_counter() - t0
        model.train()
        opt = torch.optim.Adam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
        g = torch.Generator().manual_seed(1)
        times = []
        for it in range(train_steps + 1):
            if it > 1 and time.perf_counter() - t_begin + times[-1] > budget_s:      # the next full step would overrun the safety net
                break
            sel = torch.randint(0, n_view, (train_rays if it > 0 else 32,), generator=g)
            o, d = rays_o[sel], rays_d[sel]
            target = 0.5 + 0.5 * torch.sin(3.0 * d)
            t0 = time.perf_counter()
            out = run_fixed_steps(model, o, d, num_steps, perturb=True)
            loss = F.mse_loss(out['image'], target)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            times.append(time.perf_counter() - t0)
        steps_done = len(times) - 1
        times = sorted(times[1:])
        el_t = times[len(times) // 2]
    finally:
        torch.set_num_threads(old_threads)
    cpu = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"config": "configs[0]: nerf/network.py topology (hash grid L16 C2 2^19 + nn.Linear sigma 32-64-16 + SH16 + colour 31-64-64-3), fp32, "
                      f"fixed-step renderer num_steps={num_steps}, bound {bound}, torch {torch.__version__} CPU ops",
            "kind": "port", "cores": threads, "cpu_model": cpu,
            "extrapolated": bool(done < n_view or steps_done < train_steps or train_rays != chunk),
            "protocol": f"BASELINE.md section 2: one whole {side}x{side} view in {chunk}-ray chunks; median of {train_steps} full steps of {train_rays} rays x "
                        f"{num_steps} samples after a 32-ray warm-up step; torch.set_num_threads({threads}) = the cores granted to this process. "
                        f"Reached: {done} of {n_view} rays, {steps_done} of {train_steps} steps",
            "render": {"rays_per_sec": done / el_r, "samples_per_sec": done * num_steps / el_r, "unit": "rays/s", "s_per_view": el_r * n_view / max(done, 1),
                       "sample": f"{done} rays ({-(-done // chunk)} chunks of {chunk}) of a {side}x{side} view x {num_steps} samples in {el_r:.1f} s"},
            "train": {"samples_per_sec": train_rays * num_steps / el_t, "s_per_step": el_t, "s_per_step_of_4096_rays": el_t * chunk / train_rays, "unit": "samples/s",
                      "steps_timed": steps_done,
                      "sample": f"median of {steps_done} step(s) of {train_rays} rays x {num_steps} samples (forward + backward + Adam), {el_t:.1f} s each, "
                                f"after a 32-ray warm-up step"}}
