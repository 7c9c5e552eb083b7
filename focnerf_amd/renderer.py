"""NeRFRenderer: the CALLER of the hot path, kept here so that the ops can be driven end to end where the reference tree is absent.

It offers what the reference's renderers offer to a trainer — `render`, `run` (fixed-step sampling, FOC's default: nerf/renderer.py:126-238),
`run_cuda` (occupancy-grid marching; the variant that actually runs is legacy/nerf/renderer.py:256-376),
`mark_untrained_grid`, `update_extra_state`, `reset_extra_state` — with the same arguments, buffers (`aabb_train`, `aabb_infer`,
`density_grid`, `density_bitfield`, `step_counter`) and result dictionaries, so that checkpoints and trainer code carry over; the
reference's own renderer also works unchanged on the same ops.

What differs from the reference caller:
  * the density-grid maintenance runs on the device (csrc/densitygrid.hip) instead of Python loops over 128^3 cells
    with host round trips; the torch restatements used to check it live with the tests (tests/torch_baselines.py);
  * the inference loop of `run_cuda` runs as one native call per iteration wherever that call serves the network (hash grid + the two
    FFMLPs under autocast): the list of live rays compacted on the device, 8 samples per ray and iteration, the reference's stopping
    point reproduced — same image and depth as the reference's loop, bit for bit (tests/test_gpu_network.py). `device_compaction=False`
    runs the reference's loop as it is (boolean mask, one host round trip per iteration), `=True` also takes the device-side
    compaction for networks the native call does not serve;
  * `run(..., weight_thresh=...)` names the threshold below which a sample's colour is not queried (1e-10 in FOC, 1e-4 in the legacy
    renderer), and `return_fields` controls whether the per-sample fields COMBINED.py merges are returned.
"""
import math
import os
import time

import torch
import torch.nn as nn

from . import densitygrid, raymarching

_MARCH_ALIGN = 128          # sample lists are padded to multiples of this many rows (the MLP batch granularity of the reference)


def sample_pdf(bins, weights, n_samples, det=False):
    """Inverse-CDF resampling of a piecewise-constant density (nerf/renderer.py:13-46, legacy/nerf/renderer.py:12-46): `bins` [N, M+1]
    interval edges, `weights` [N, M] -> `n_samples` positions per row, at the centres of equal-probability strata (`det`) or at uniform
    random quantiles. Torch ops in the reference's order (cumsum, searchsorted right=True, gather, linear interpolation inside the
    interval, intervals of mass < 1e-5 treated as unit mass)."""
    weights = weights + 1e-5
    cdf = torch.cumsum(weights / torch.sum(weights, -1, keepdim=True), -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if det:
        u = torch.linspace(0. + 0.5 / n_samples, 1. - 0.5 / n_samples, steps=n_samples).to(weights.device)
        u = u.expand(list(cdf.shape[:-1]) + [n_samples])
    else:
        u = torch.rand(list(cdf.shape[:-1]) + [n_samples]).to(weights.device)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.max(torch.zeros_like(inds - 1), inds - 1)
    above = torch.min((cdf.shape[-1] - 1) * torch.ones_like(inds), inds)
    pair = torch.stack([below, above], -1)
    shape = [pair.shape[0], pair.shape[1], cdf.shape[-1]]
    cdf_g = torch.gather(cdf.unsqueeze(1).expand(shape), 2, pair)
    bins_g = torch.gather(bins.unsqueeze(1).expand(shape), 2, pair)
    mass = cdf_g[..., 1] - cdf_g[..., 0]
    mass = torch.where(mass < 1e-5, torch.ones_like(mass), mass)
    return bins_g[..., 0] + (u - cdf_g[..., 0]) / mass * (bins_g[..., 1] - bins_g[..., 0])


_SIDE_STREAMS = {}


def _side_streams(dev, n):
    """n side streams per device for the staged render (made once: stream creation is not free, and scratch buffers are keyed by stream)."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), n)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = [torch.cuda.Stream(device=dev) for _ in range(n)]
    return _SIDE_STREAMS[key]


def _flat_rays(rays_o, rays_d):
    lead = tuple(rays_o.shape[:-1])
    return rays_o.contiguous().view(-1, 3), rays_d.contiguous().view(-1, 3), lead


class NeRFRenderer(nn.Module):
    def __init__(self, bound=1, cuda_ray=False, density_scale=1, min_near=0.2, density_thresh=0.01, bg_radius=-1):
        super().__init__()
        self.bound, self.density_scale, self.min_near = bound, density_scale, min_near
        self.density_thresh, self.bg_radius, self.cuda_ray = density_thresh, bg_radius, cuda_ray
        self.cascade = 1 + math.ceil(math.log2(bound))          # one occupancy cascade per doubling of the box
        self.grid_size = 128
        box = torch.tensor([-bound] * 3 + [bound] * 3, dtype=torch.float32)
        self.register_buffer('aabb_train', box)
        self.register_buffer('aabb_infer', box.clone())
        if cuda_ray:
            cells = self.grid_size ** 3
            self.register_buffer('density_grid', torch.zeros(self.cascade, cells))
            self.register_buffer('density_bitfield', torch.zeros(self.cascade * cells // 8, dtype=torch.uint8))
            self.register_buffer('step_counter', torch.zeros(16, 2, dtype=torch.int32))      # samples marched in the last 16 training steps
            self._mean_density, self._mean_density_dev = 0, None
            self.iter_density = self.mean_count = self.local_step = 0

    # the reference keeps `mean_density` as a Python float; the device-side update leaves it on the GPU until somebody asks
    @property
    def mean_density(self):
        if getattr(self, "_mean_density_dev", None) is not None:
            self._mean_density, self._mean_density_dev = float(self._mean_density_dev.item()), None
        return self._mean_density

    @mean_density.setter
    def mean_density(self, value):
        self._mean_density, self._mean_density_dev = value, None

    # ---- what a network has to provide
    def forward(self, x, d):
        raise NotImplementedError()

    def density(self, x):
        raise NotImplementedError()

    def color(self, x, d, mask=None, **kwargs):
        raise NotImplementedError()

    def _aabb(self):
        return self.aabb_train if self.training else self.aabb_infer

    def _background_colour(self, rays_o, rays_d, bg_color):
        if self.bg_radius > 0:
            return self.background(raymarching.sph_from_ray(rays_o, rays_d, self.bg_radius), rays_d)
        return 1 if bg_color is None else bg_color

    # ------------------------------------------------------------------ fixed number of samples per ray
    def run(self, rays_o, rays_d, yolo_details=None, num_steps=512, upsample_steps=0, bg_color=None, perturb=False, weight_thresh=1e-10,
            return_fields=None, **kwargs):
        """`num_steps` equidistant samples between the box entry and exit of every ray, density at all of them, colour where the
        compositing weight exceeds `weight_thresh`, then alpha compositing with torch ops — the computation of nerf/renderer.py:126-238
        for `upsample_steps=0` (the only value FOC uses, main_nerf.py:31-32), in its order of operations.

        yolo_details = (sample mask [1,N,T], box, object feature) from the trainer (nerf/utils.py:57-154): handed on to `color()`, and
        in training it yields `criterion_outside_mask`, the norm of the densities outside the mask (:163-165).
        Result: depth, image, weights_sum, criterion_outside_mask, timing (host seconds before / after the colour query) and, with
        `return_fields` (default: only in eval mode), densities [N,T,1] and rgbs [N,T,3]."""
        want_fields = (not self.training) if return_fields is None else return_fields
        started = time.time()
        o, d, lead = _flat_rays(rays_o, rays_d)
        n, T, box = o.shape[0], num_steps, self._aabb()

        near, far = (t.unsqueeze(-1) for t in raymarching.near_far_from_aabb(o, d, box, self.min_near))
        span = far - near
        spacing = span / T
        z = near + span * torch.linspace(0.0, 1.0, T, device=o.device).unsqueeze(0).expand(n, T)
        if perturb:
            z = z + (torch.rand(z.shape, device=o.device) - 0.5) * spacing
        points = o.unsqueeze(-2) + d.unsqueeze(-2) * z.unsqueeze(-1)
        points = torch.min(torch.max(points, box[:3]), box[3:])

        field = self.density(points.reshape(-1, 3))
        sigma = field['sigma'].view(n, T)
        outside = None
        if self.training and yolo_details is not None:
            outside = torch.norm(sigma[~yolo_details[0].squeeze(0)] - 0)
        if upsample_steps > 0:
            # hierarchical resampling, legacy/nerf/renderer.py:169-202 (COMBINED.py:485-514 carries the same branch; FOC's own run() dropped
            # it and FOC runs upsample_steps = 0): weights of the coarse pass -> `upsample_steps` more samples per ray where they are large
            # (sample_pdf; deterministic outside training), density at the new points only, both sets merged in depth order
            t_new = int(upsample_steps)
            with torch.no_grad():
                step0 = torch.cat([z[..., 1:] - z[..., :-1], spacing * torch.ones_like(z[..., :1])], dim=-1)
                alpha0 = 1 - torch.exp(-step0 * self.density_scale * sigma)
                w0 = alpha0 * torch.cumprod(torch.cat([torch.ones_like(alpha0[..., :1]), 1 - alpha0 + 1e-15], dim=-1), dim=-1)[..., :-1]
                mid = z[..., :-1] + 0.5 * step0[..., :-1]
                z_new = sample_pdf(mid, w0[:, 1:-1], t_new, det=not self.training).detach()
                points_new = o.unsqueeze(-2) + d.unsqueeze(-2) * z_new.unsqueeze(-1)
                points_new = torch.min(torch.max(points_new, box[:3]), box[3:])
            field_new = self.density(points_new.reshape(-1, 3))
            z, order = torch.sort(torch.cat([z, z_new], dim=1), dim=1)
            points = torch.gather(torch.cat([points, points_new], dim=1), 1, order.unsqueeze(-1).expand(n, T + t_new, 3))
            merged = {}
            for k in field:
                both = torch.cat([field[k].view(n, T, -1), field_new[k].view(n, t_new, -1)], dim=1)
                merged[k] = torch.gather(both, 1, order.unsqueeze(-1).expand_as(both))
            field = merged
            T = T + t_new
            sigma = field['sigma'].view(n, T)

        step = torch.cat([z[..., 1:] - z[..., :-1], spacing * torch.ones_like(z[..., :1])], dim=-1)
        alpha = 1 - torch.exp(-step * self.density_scale * sigma)
        survive = torch.cat([torch.ones_like(alpha[..., :1]), 1 - alpha + 1e-15], dim=-1)
        weights = alpha * torch.cumprod(survive, dim=-1)[..., :-1]

        per_sample = {k: v.reshape(n * T, -1) for k, v in field.items()}
        view_dirs = d.view(-1, 1, 3).expand_as(points)
        colour = self.color(points.reshape(-1, 3), view_dirs.reshape(-1, 3), mask=(weights > weight_thresh).reshape(-1),
                            yolo_details=yolo_details, **per_sample).view(n, -1, 3)
        queried = time.time()

        opacity = weights.sum(dim=-1)
        depth = torch.sum(weights * ((z - near) / span).clamp(0, 1), dim=-1)
        image = torch.sum(weights.unsqueeze(-1) * colour, dim=-2)
        image = image + (1 - opacity).unsqueeze(-1) * self._background_colour(o, d, bg_color)

        out = {'depth': depth.view(*lead), 'image': image.view(*lead, 3), 'weights_sum': opacity, 'criterion_outside_mask': outside,
               'timing': [queried - started, time.time() - queried]}
        if want_fields:
            out['densities'], out['rgbs'] = sigma.unsqueeze(-1), colour
        return out

    # ------------------------------------------------------------------ occupancy-grid marching
    def _finish(self, image, depth, opacity, near, far, background, lead):
        rest = (1 - opacity).unsqueeze(-1)
        # (1 - w) * 1 is (1 - w) bit for bit: the default white background needs no multiply (one launch less, forward and backward)
        image = image + (rest if isinstance(background, (int, float)) and background == 1 else rest * background)
        with torch.no_grad():                                       # the depth carries no gradient (raymarching.py:275): keep it out of the graph
            depth = torch.clamp(depth - near, min=0) / (far - near)
        return image.view(*lead, 3), depth.view(*lead)

    def _scaled(self, sigmas):
        return sigmas if self.density_scale == 1 else self.density_scale * sigmas      # x * 1 == x: no launch for the default scale

    def run_cuda(self, rays_o, rays_d, dt_gamma=0, bg_color=None, perturb=False, force_all_rays=False, max_steps=1024, T_thresh=1e-4,
                 device_compaction=None, **kwargs):
        """Samples only where the occupancy bitfield is set (legacy/nerf/renderer.py:256-376). Training: one marching pass, one
        evaluation of the field, one compositing node. Inference: rays advance a few samples at a time and leave the list once they are
        opaque or out of the box."""
        o, d, lead = _flat_rays(rays_o, rays_d)
        n, dev = o.shape[0], o.device
        out = {}
        if self.training:
            from .occtrain import occ_train_fusable, render_occupancy_train
            if o.is_cuda and torch.is_grad_enabled() and torch.is_autocast_enabled() and occ_train_fusable(self):
                # the whole training forward as one autograd node (focnerf_amd/occtrain.py): same samples, same image; the box test of
                # near_far_from_aabb rides in the march's count pass
                slot = self.step_counter[self.local_step % 16]
                slot.zero_()
                self.local_step += 1
                image, opacity, depth = render_occupancy_train(self, o.float(), d.float(), slot, bg_color, perturb, force_all_rays, dt_gamma, max_steps, T_thresh,
                                                               _MARCH_ALIGN)
                out['weights_sum'] = opacity
                out['image'], out['depth'] = image.view(*lead, 3), depth.view(*lead)
                return out
        near, far = raymarching.near_far_from_aabb(o, d, self._aabb(), self.min_near)
        background = self._background_colour(o, d, bg_color)
        if self.training:
            slot = self.step_counter[self.local_step % 16]
            slot.zero_()
            self.local_step += 1
            xyzs, dirs, deltas, rays = raymarching.march_rays_train(o, d, self.bound, self.density_bitfield, self.cascade, self.grid_size, near,
                                                                    far, slot, self.mean_count, perturb, _MARCH_ALIGN, force_all_rays,
                                                                    dt_gamma, max_steps)
            sigmas, rgbs = self(xyzs, dirs)
            opacity, depth, image = raymarching.composite_rays_train(self._scaled(sigmas), rgbs, deltas, rays, T_thresh)
            out['weights_sum'] = opacity
        else:
            opacity, depth, image = (torch.zeros(n, *tail, dtype=torch.float32, device=dev) for tail in ((), (), (3,)))
            alive = torch.arange(n, dtype=torch.int32, device=dev)
            t_now = near.clone()
            marched = 0
            # device_compaction: None (default) = the native loop where it serves the network (same image and depth as the reference's loop,
            # bit for bit: tests/test_gpu_network.py), else the reference's loop as it is; True = the native loop, else the Python loop with the
            # list compacted on the device and its length read late; False = the reference's loop (boolean mask, a host round trip per iteration)
            if (device_compaction is None or device_compaction) and self._native_loop_ok(o):
                self._native_inference_loop(o, d, near, far, alive, t_now, opacity, depth, image, perturb, dt_gamma, max_steps, T_thresh)
                marched = max_steps                                   # the Python loop below has nothing left to do
            device_compaction = bool(device_compaction)
            import contextlib
            from .field import half_cache_scope
            # the loop evaluates the same parameters once per burst: one fp16 conversion per view when nothing can write them in between
            # device_compaction: the list of live rays is compacted on the device, and its length is read back LATE — the loop sizes
            # iteration i by the count of iteration i - 1 - lag (an upper bound: rays only die), entries behind the true count are -1 and the
            # kernels skip them. With the reference's `rays_alive[rays_alive >= 0]` (or an immediate `.item()`) the host waits for the GPU
            # once per iteration and the GPU then waits for the host to enqueue the next one: 0.54 ms per iteration for 0.17 ms of kernels.
            # A stale count can only give a SHORTER burst than the reference's rule would; every ray receives the same samples in the same
            # order, but rays that are still alive after max_steps samples stop where THIS loop's `marched` passes max_steps, which can be a
            # few samples (< 8) away from where the reference's would. The native loop below reproduces the reference's stopping point
            # exactly; this Python form is the fallback for networks the native step does not serve.
            lag = int(os.environ.get("FOC_RENDER_COUNT_LAG", "2")) if device_compaction else 0
            ring = torch.empty(lag + 1, dtype=torch.int32).pin_memory() if lag > 0 else None
            waiting = []                                              # (event, ring slot) of counts on their way to the host
            it = 0
            still = torch.zeros(n, dtype=torch.float32, device=dev)   # "no jitter" for every iteration after the first: one fill per view
            with (half_cache_scope() if not torch.is_grad_enabled() else contextlib.nullcontext()):
                while marched < max_steps and alive.shape[0] > 0:
                    live = alive.shape[0]
                    burst = max(min(n // live, 8), 1)                 # fewer live rays -> more samples per ray and launch
                    xyzs, dirs, deltas = raymarching.march_rays(live, burst, alive, t_now, o, d, self.bound, self.density_bitfield, self.cascade,
                                                                self.grid_size, near, far, _MARCH_ALIGN, perturb and marched == 0, dt_gamma,
                                                                max_steps, noises=None if (perturb and marched == 0) else still)
                    sigmas, rgbs = self(xyzs, dirs)
                    raymarching.composite_rays(live, burst, alive, t_now, self._scaled(sigmas), rgbs, deltas, opacity, depth, image, T_thresh)
                    if device_compaction and lag > 0:
                        kept, count = raymarching.compact_alive(alive, pad=True)
                        slot = it % (lag + 1)
                        ring[slot:slot + 1].copy_(count, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record()
                        waiting.append((ev, slot))
                        if len(waiting) > lag:                        # the count of `lag` iterations ago: long since on the host
                            ev0, s0 = waiting.pop(0)
                            ev0.synchronize()
                            live = min(live, int(ring[s0]))
                        alive = kept[:live]
                    elif device_compaction:
                        kept, count = raymarching.compact_alive(alive)
                        alive = kept[:int(count.item())]
                    else:
                        alive = alive[alive >= 0]
                    marched += burst
                    it += 1
        out['image'], out['depth'] = self._finish(image, depth, opacity, near, far, background, lead)
        return out

    # ------------------------------------------------------------------ the inference loop, one native call per iteration
    def _native_loop_ok(self, o):
        """`foc_occ_render_step` serves the networks of the whole-field kernel (hash grid D = 3, C = 2 -> 64-wide sigma net -> SH + 64-wide
        colour net) at density_scale 1, on the GPU, under autocast (fp16 table and weights)."""
        from .field import infer_fusable
        return (o.is_cuda and torch.is_autocast_enabled() and not torch.is_grad_enabled() and infer_fusable(self) and self.density_scale == 1
                and not getattr(self, "uses_object_feature", False) and self.encoder.gridtype_id == 0 and not self.encoder.align_corners
                and self.encoder.interp_id == 0 and os.environ.get("FOC_RENDER_NATIVE", "1") != "0")

    def _native_inference_loop(self, o, d, near, far, alive, t_now, opacity, depth, image, perturb, dt_gamma, max_steps, T_thresh):
        """The loop of legacy/nerf/renderer.py:323-372 with every iteration ONE call into the library (csrc/occrender.hip: march, encode,
        whole-field kernel, composite, compaction) on buffers allocated once per view; bursts of FOC_RENDER_BURST samples per ray, the live
        count read `FOC_RENDER_COUNT_LAG` iterations late, the reference's stopping point reproduced. Same samples, same per-ray
        accumulation order: the same image and depth bit for bit on every configuration the tests run (jittered first samples with the
        same noise included), with one caveat spelled out at the burst rule below — a re-derivation of t that can differ by an ulp where a
        single advance more than doubles t while fewer than half of the rays are alive."""
        import numpy as np
        from ._lib import lib, ptr, stream_of, check
        from .field import _half_of, half_cache_scope
        n, dev = o.shape[0], o.device
        enc, sn, cn = self.encoder, self.sigma_net, self.color_net
        L = enc.offsets.shape[0] - 1
        lag = max(1, int(os.environ.get("FOC_RENDER_COUNT_LAG", "2")))
        # Samples per ray and iteration. The reference sizes a burst so that live x burst stays within the view's ray count (max(min(n // live,
        # 8), 1), renderer.py:337: ONE sample per ray for the ~130 iterations in which most rays are alive) — a memory bound of its time. What
        # a ray receives does not depend on how its samples are dealt over iterations, provided the march continues after each sample from
        # the t composite_rays would hand to the next iteration (rays_t + deltas[:,1]: `flags` below) — composite_rays accumulates sample by
        # sample anyway. So an iteration here marches at least FOC_RENDER_BURST samples per ray (default 8, at most 16; 1 = the reference's
        # schedule): 8x fewer iterations, the rays' state read and written once per 8 samples. A ray that dies inside a burst wastes the rest
        # of it, as it does in the reference's own bursts of 8.
        # What DOES depend on the schedule is where the reference's loop stops for rays that are still alive after max_steps samples: `step +=
        # n_step` overshoots max_steps by up to 7, depending on how many rays were alive at each of its iterations. The wide bursts therefore
        # end 8 samples short of max_steps; if rays are still alive there, the reference's own schedule is replayed from the histogram of the
        # sample index at which every ray died (`deaths`, filled by the composite kernel) and the last iterations run exactly as the reference
        # would run them (its burst rule on the exact live count, one host round trip each) — same image, bit for bit, cap included.
        wide = min(max(int(os.environ.get("FOC_RENDER_BURST", "8")), 1), 16)
        cap = n * wide                                         # most samples of one iteration (the reference's own bursts keep live x burst <= n)
        from ._lib import get_option
        piece = min(cap, max(1024, get_option("FOC_OCC_FIELD_PIECE")))
        samples = torch.empty(cap * 8, dtype=torch.float32, device=dev)
        planes = torch.empty(L * piece * 2, dtype=torch.float16, device=dev)
        sigma, rgb = torch.empty(cap, dtype=torch.float32, device=dev), torch.empty(cap * 3, dtype=torch.float32, device=dev)
        lists = [alive, torch.empty_like(alive)]
        count = torch.empty(1, dtype=torch.int32, device=dev)
        scratch = torch.empty(lib.foc_occ_render_step_scratch_bytes(n), dtype=torch.uint8, device=dev)
        still = torch.zeros(n, dtype=torch.float32, device=dev)
        jitter = torch.rand(n, dtype=torch.float32, device=dev) if perturb else still
        ring = torch.empty(lag + 1, dtype=torch.int32).pin_memory()
        n_deaths = int(max_steps) + 32
        deaths = torch.zeros(n_deaths, 64, dtype=torch.int32, device=dev)       # [sample index][slice]: see RM_DEATH_SLICES
        trace = bool(os.environ.get("FOC_RENDER_TRACE"))
        S = float(np.log2(enc.per_level_scale))

        def rule(n_alive):                                      # renderer.py:337
            return max(min(n // n_alive, 8), 1)

        with half_cache_scope():
            emb, ws, wc = _half_of(enc.embeddings), _half_of(sn.weights), _half_of(cn.weights)
            st = stream_of(o)
            state = {"it": 0, "marched": 0}

            def step(live, burst, flags):
                it, marched = state["it"], state["marched"]
                if trace:
                    print(f"[occ loop] it {it} live<= {live} burst {burst} flags {flags} marched {marched}", flush=True)
                src, dst = lists[it & 1], lists[(it & 1) ^ 1]
                check(lib.foc_occ_render_step(live, burst, ptr(src), ptr(dst), ptr(count), ptr(t_now), ptr(o), ptr(d), float(self.bound), float(dt_gamma),
                                              int(max_steps), self.cascade, self.grid_size, ptr(self.density_bitfield), ptr(near), ptr(far),
                                              ptr(jitter if marched == 0 else still), ptr(samples), ptr(planes), ptr(sigma), ptr(rgb), ptr(emb),
                                              ptr(enc.offsets), None, L, S, enc.base_resolution, ptr(ws), sn.num_layers, ptr(wc), cn.num_layers, sn.activation,
                                              None, float(T_thresh), ptr(opacity), ptr(depth), ptr(image), ptr(scratch), flags, ptr(deaths), marched, n_deaths,
                                              st), "occ_render_step")
                state["it"], state["marched"] = it + 1, marched + burst

            # ---- the wide bursts, the live count read `lag` iterations late (an upper bound: rays only die)
            wide_end = max(int(max_steps) - 8, 0) if wide > 1 else 0
            waiting, live = [], n
            while state["marched"] < wide_end and live > 0:
                ref_burst = rule(live)
                burst = max(ref_burst, min(wide, wide_end - state["marched"]))
                if perturb and state["marched"] == 0:
                    # The jitter belongs to the samples of the reference's FIRST iteration only: its march offsets the start by noise * dt, but
                    # composite_rays advances rays_t from the UN-jittered start by the deltas (raymarching.cu:736-748, :868-905), so the second
                    # iteration continues without the offset. A wide first burst would carry it through all of its samples: the first
                    # iteration keeps the reference's own length.
                    burst = ref_burst
                # several of the reference's one-sample iterations in one: every sample starts from the t the reference's NEXT iteration would
                # start from, last_t + (t - last_t) as composite_rays re-derives it (an ulp off the march's own t when the subtraction rounds).
                # Where the reference's own bursts are 2..7 samples long (fewer than half of the rays alive) it re-derives at ITS iteration
                # boundaries, which depend on the exact live count of every iteration — not known here without a host round trip per
                # iteration. The two differ only where fl(t - last_t) is inexact, i.e. t > 2 last_t: one advance that more than doubles t.
                # That happens at a ray's first samples (camera inside the box, near = min_near, a long empty stretch) — the phase in which
                # every ray is alive, the reference's burst is 1 and this loop re-derives after every sample — and not later in a
                # single-object scene, where a ray that has reached the object at t >= 1 finds nothing beyond it at 2 t.
                step(live, burst, 1 if (ref_burst == 1 and burst > 1) else 0)
                slot = (state["it"] - 1) % (lag + 1)
                ring[slot:slot + 1].copy_(count, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                waiting.append((ev, slot))
                if len(waiting) > lag:
                    ev0, s0 = waiting.pop(0)
                    ev0.synchronize()
                    live = min(live, int(ring[s0]))
            if live <= 0:
                return
            # ---- up to max_steps (and the reference's overshoot): the reference's own iterations
            if state["it"] > 0:
                live = int(count.item())                        # exact from here on
            if live <= 0:
                return
            done = state["marched"]
            virt = 0                                            # the reference's `step` at its last iteration boundary <= done
            if done > 0:
                died = torch.cumsum(deaths.sum(1), 0).cpu().tolist()   # died[i] = rays whose last sample index is <= i
                while True:
                    b = rule(n - (died[virt - 1] if virt > 0 else 0))
                    if virt + b > done:
                        break
                    virt += b
                nxt = virt + b
            else:
                nxt = rule(live)
            while True:
                step(live, nxt - state["marched"], 0)
                live = int(count.item())
                if state["marched"] >= max_steps or live <= 0:
                    break
                nxt = state["marched"] + rule(live)

    # ------------------------------------------------------------------ occupancy-grid maintenance (device side)
    def _require_grid(self, what):
        if not self.density_bitfield.is_cuda:
            raise RuntimeError(f"{what}: the occupancy grid is maintained by HIP kernels; move the module to the GPU first")

    def reset_extra_state(self):
        if not self.cuda_ray:
            return
        self.density_grid.zero_()
        self.step_counter.zero_()
        self.mean_density = 0
        self.iter_density = self.mean_count = self.local_step = 0

    @torch.no_grad()
    def mark_untrained_grid(self, poses, intrinsic, S=64):
        """Cells that no training camera sees get density -1: they are never marched and never updated (nerf/renderer.py:356-418).
        Returns the per-cell camera count [cascade, H^3]. (`S`, the reference's chunk size, has no meaning here.)"""
        if not self.cuda_ray:
            return None
        self._require_grid("mark_untrained_grid")
        poses = torch.as_tensor(poses).to(self.density_bitfield.device).float()
        return densitygrid.mark_untrained_grid(poses, intrinsic, self.bound, self.cascade, self.grid_size, self.density_grid, return_count=True)

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128):
        """Re-estimates the occupancy grid from the current density field (legacy/nerf/renderer.py:445-536): the first 16 calls visit
        every cell, later calls visit H^3/4 random cells plus H^3/4 random occupied cells per cascade; the grid keeps
        max(old * decay, new), and cells above min(mean density, density_thresh) are marked occupied. Also refreshes `mean_count`,
        the sample budget of the next training steps, from the last (up to) 16 marching passes — the one host read of this call."""
        if not self.cuda_ray:
            return
        self._require_grid("update_extra_state")
        dev, C, H = self.density_bitfield.device, self.cascade, self.grid_size
        if self.iter_density < 16:
            visited = None
            at = densitygrid.grid_cells_xyz(C, H, self.bound, torch.rand(C * H ** 3, 3, device=dev), dev)
        else:
            k = H ** 3 // 4
            visited, at = densitygrid.grid_update_sample(self.density_grid, C, H, self.bound,
                                                         torch.randint(0, H, (C, k, 3), device=dev, dtype=torch.int32),
                                                         torch.rand(C, k, device=dev), torch.rand(C * 2 * k, 3, device=dev))
        sigmas = self.density(at)['sigma'].reshape(-1).detach()
        mean = torch.empty(1, dtype=torch.float32, device=dev)
        densitygrid.grid_update_apply(self.density_grid, C, H, sigmas, visited, self.density_scale, decay, self.density_thresh,
                                      self.density_bitfield, mean)
        self._mean_density_dev = mean
        self.iter_density += 1
        recent = min(16, self.local_step)
        if recent > 0:
            self.mean_count = int(self.step_counter[:recent, 0].sum().item() / recent)
        self.local_step = 0

    @torch.no_grad()
    def set_density_grid(self, density_grid, density_thresh=None):
        """Installs a given grid [cascade, H^3] (Morton order) and packs it: the synthetic scenes of bench.py and the tests use it in
        place of a trained grid."""
        assert self.cuda_ray
        self.density_grid.copy_(density_grid)
        self.mean_density = torch.mean(self.density_grid.clamp(min=0)).item()
        level = min(self.mean_density, self.density_thresh) if density_thresh is None else density_thresh
        self.density_bitfield = raymarching.packbits(self.density_grid, level, self.density_bitfield)

    # ------------------------------------------------------------------ whole views
    def render(self, rays_o, rays_d, yolo_details=None, staged=False, max_ray_batch=4096, **kwargs):
        """rays [B,N,3] -> result dictionary of `run` / `run_cuda`. `staged` (fixed-step path only) walks a view in chunks of
        `max_ray_batch` rays and assembles depth [B,N], image [B,N,3] and — when the chunks carry them — densities [B,N,T] and
        rgbs [B,N,T,3] for the whole view, as nerf/renderer.py:511-560 does (1.3 + 3.9 GB for 800 x 800 x 512)."""
        if self.cuda_ray:
            return self.run_cuda(rays_o, rays_d, **kwargs)                    # the marching path knows no yolo_details (renderer.py:243)
        if not staged:
            return self.run(rays_o, rays_d, yolo_details, **kwargs)
        B, N = rays_o.shape[:2]
        dev = rays_o.device
        depth, image = torch.empty(B, N, device=dev), torch.empty(B, N, 3, device=dev)
        densities = rgbs = None
        want_fields = kwargs.get("return_fields")
        if (not self.training) if want_fields is None else want_fields:
            # samples per ray of the dispatched run(): its own defaults where the caller passed none (the view buffers are made before
            # the first chunk so that a fused path can write into them in place; a chunk that returns another T raises below)
            import inspect
            params = inspect.signature(type(self).run).parameters
            defaults = {k: params[k].default for k in ("num_steps", "upsample_steps") if k in params and params[k].default is not inspect.Parameter.empty}
            base = inspect.signature(NeRFRenderer.run).parameters
            T = int(kwargs.get("num_steps", defaults.get("num_steps", base["num_steps"].default))) + \
                int(kwargs.get("upsample_steps", defaults.get("upsample_steps", base["upsample_steps"].default)))
            densities, rgbs = torch.empty(B, N, T, device=dev), torch.empty(B, N, T, 3, device=dev)
        import contextlib
        from .field import half_cache_scope
        # Inference of whole views without per-sample outputs: the rays are walked in 8 x 8 pixel tiles (rayorder.py) and the image rows put
        # back where the caller's rays were — a ray's result does not depend on its neighbours in a chunk (no jitter: perturb must be off)
        perms = None
        per_ray_bg = torch.is_tensor(kwargs.get("bg_color")) and kwargs["bg_color"].numel() > 3
        if (densities is None and not torch.is_grad_enabled() and not self.training and not kwargs.get("perturb", False) and not per_ray_bg
                and dev.type == "cuda"):
            from .rayorder import view_tiling
            perms = [view_tiling(rays_d[b]) for b in range(B)]
            if any(p is None for p in perms):
                perms = None
        if perms is not None:
            caller_depth, caller_image = depth, image
            rays_o = torch.stack([rays_o[b].index_select(0, perms[b]) for b in range(B)]) if B > 1 else rays_o[0].index_select(0, perms[0]).unsqueeze(0)
            rays_d = torch.stack([rays_d[b].index_select(0, perms[b]) for b in range(B)]) if B > 1 else rays_d[0].index_select(0, perms[0]).unsqueeze(0)
            depth, image = torch.empty_like(depth), torch.empty_like(image)
        # one fp16 conversion of the table / weight blobs per VIEW instead of per chunk — only where nothing can write the parameters
        # between two chunks (no autograd, hence no optimizer step inside the loop); the copies are dropped when the view is done
        # Chunks are independent of each other while the kernels of one chunk form a chain: under no_grad the chunks alternate between
        # two side streams, so that the launch gaps and the narrow kernels (near/far, sampling, compositing) of one chunk run under the
        # wide kernels of the next (FOC_RENDER_STREAMS=1: everything on the caller's stream).
        n_streams = int(os.environ.get("FOC_RENDER_STREAMS", "2")) if (not torch.is_grad_enabled() and dev.type == "cuda") else 1
        main = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
        with (half_cache_scope() if not torch.is_grad_enabled() else contextlib.nullcontext()):
            sides = []
            if n_streams > 1:
                self._warm_half_cache()                                       # the fp16 copies are made on the caller's stream, before the fork
                sides = _side_streams(dev, n_streams)
                for st in sides:
                    st.wait_stream(main)
                    # the view's buffers were allocated on the caller's stream and are written on the side streams: the caching allocator must
                    # not hand their memory out again before the side streams' work on them is done, whatever path leaves this function
                    for buf in (depth, image, densities, rgbs):
                        if buf is not None:
                            buf.record_stream(st)
            chunk = 0
            # (gating the encoder launches of the two streams behind each other — so that an encoder only ever runs next to a whole-field kernel,
            # never next to another encoder — measured 42.6 against 41.6 ms per view: the free interleaving is the better one)
            # `max_ray_batch` bounds what the caller's memory has to hold at once (4096 rays in the reference's flags: a few hundred MB on
            # a 24 GB card). The fused inference path writes straight into the view's buffers and a ray's result does not depend on its
            # chunk, so it walks the view in pieces of at least 16384 rays (0.6 GB of encoder planes): fewer, larger launches — 42.6 ->
            # 40.9 ms per 800 x 800 view (8192: 41.6, 32768: 42.5, 65536: 45.2; FOC_RENDER_MIN_CHUNK=0: the caller's chunks as they are)
            piece = max_ray_batch
            if not torch.is_grad_enabled() and dev.type == "cuda" and kwargs.get("fused"):
                from .field import infer_fusable
                if infer_fusable(self):
                    piece = max(max_ray_batch, int(os.environ.get("FOC_RENDER_MIN_CHUNK", "16384")))
            try:
                for b in range(B):
                    for lo in range(0, N, piece):
                        hi = min(lo + piece, N)
                        # a fused path may write straight into the view's buffers (`_out`); anything else is copied in
                        into = (depth[b, lo:hi], image[b, lo:hi]) + ((densities[b, lo:hi], rgbs[b, lo:hi]) if densities is not None else ())
                        with (torch.cuda.stream(sides[chunk % n_streams]) if sides else contextlib.nullcontext()):
                            part = self._render_chunk(rays_o, rays_d, b, lo, hi, yolo_details, into, kwargs)
                        chunk += 1
            finally:
                for st in sides:                                              # also when a chunk raised: the caller's stream must not run ahead of
                    main.wait_stream(st)                                      # side-stream kernels that still write the view's buffers
        if perms is not None:
            for b in range(B):
                caller_depth[b].index_copy_(0, perms[b], depth[b])
                caller_image[b].index_copy_(0, perms[b], image[b])
            depth, image = caller_depth, caller_image
        out = {'depth': depth, 'image': image, 'timing': part.get('timing')}
        if densities is not None and 'densities' in part:
            out['densities'], out['rgbs'] = densities, rgbs
        return out

    def _render_chunk(self, rays_o, rays_d, b, lo, hi, yolo_details, into, kwargs):
        """One chunk of a staged render: `run` on rays [lo, hi) of view b; whatever it did not write in place goes into `into` =
        (depth, image[, densities, rgbs]) views of the whole-view buffers."""
        part = self.run(rays_o[b:b + 1, lo:hi], rays_d[b:b + 1, lo:hi], yolo_details, _out=into, **kwargs)
        if part['depth'].data_ptr() != into[0].data_ptr():
            into[0].copy_(part['depth'].view(into[0].shape))
        if part['image'].data_ptr() != into[1].data_ptr():
            into[1].copy_(part['image'].view(into[1].shape))
        if 'densities' in part and len(into) == 4:
            if part['densities'].numel() != into[2].numel():
                raise RuntimeError(f"staged render: run() returned {part['densities'].numel() // max(1, into[0].numel())} samples per ray, the view buffers "
                                   f"hold {into[2].numel() // max(1, into[0].numel())} (pass num_steps / upsample_steps to render())")
            if part['densities'].data_ptr() != into[2].data_ptr():
                into[2].copy_(part['densities'].view(into[2].shape))
            if part['rgbs'].data_ptr() != into[3].data_ptr():
                into[3].copy_(part['rgbs'].view(into[3].shape))
        return part

    def _warm_half_cache(self):
        from .field import _half_of
        for mod in (getattr(self, "encoder", None), getattr(self, "sigma_net", None), getattr(self, "color_net", None)):
            for name in ("embeddings", "weights"):
                p = getattr(mod, name, None) if mod is not None else None
                if torch.is_tensor(p):
                    _half_of(p)
