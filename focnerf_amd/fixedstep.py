"""Fused fixed-step render path (FOC's default: num_steps samples per ray, no occupancy grid).

`render_fixed_steps(model, rays_o, rays_d, ...)` computes what `NeRFRenderer.run` computes
(nerf/renderer.py:126-238 with upsample_steps=0) for a `focnerf_amd.network.NeRFNetwork`, with the torch
glue replaced by the kernels of csrc/fixedstep.hip:

    near_far_from_aabb -> fixed_sample -> grid_encode -> sigma_net -> [density head] -> color_net -> [composite]

The two bracketed stages are autograd Functions defined here. The colour network is evaluated on every
sample and masked by `weights > weight_thresh` inside the composite, which yields the same image and the
same gradients as the reference's gather -> MLP -> scatter (masked-out samples get rgb = 0 and no gradient).
SURVEY.md §8(f)-3. Parity is tested against `NeRFRenderer.run` on the same network (tests/test_gpu_fixedstep.py).
"""
import torch
from torch.autograd import Function

from . import raymarching
from ._lib import lib, ptr, stream_of, check
from .gridencoder import grid_encode


RAY_BLOCK = 64      # block-interleaved sample order of the inference path (csrc/fixedstep.hip, fs_block_row)


def ray_block_default():
    """Sample order of the fused inference path: 64-ray blocks unless FOC_RAY_BLOCK=0 (ray-major, the round-1 order)."""
    import os
    return RAY_BLOCK if os.environ.get("FOC_RAY_BLOCK", str(RAY_BLOCK)) != "0" else 0


def blocked_rows(N, T, ray_block):
    """Rows of the per-sample arrays of N rays x T samples in the given order (the last 64-ray block is padded)."""
    return N * T if not ray_block else -(-N // ray_block) * ray_block * T


def fixed_sample(rays_o, rays_d, nears, fars, aabb, noise, T, bound, want_xyzs=False, want_ray_sh=False, ray_block=0):
    """ray_block = 64: rows in the block-interleaved order (include/focnerf.h), `blocked_rows(N, T, 64)` of them."""
    N = rays_o.shape[0]
    dev = rays_o.device
    rows = blocked_rows(N, T, ray_block)
    enc_in = torch.empty(rows, 3, dtype=torch.float32, device=dev)
    xyzs = torch.empty(rows, 3, dtype=torch.float32, device=dev) if want_xyzs else None
    ray_sh = torch.empty(N, 16, dtype=torch.float16, device=dev) if want_ray_sh else None
    check(lib.foc_fixed_sample(ptr(rays_o), ptr(rays_d), ptr(nears), ptr(fars), ptr(aabb), ptr(noise), N, T, float(bound), ptr(xyzs), ptr(enc_in),
                               ptr(ray_sh), int(ray_block), stream_of(rays_o)), "fixed_sample")
    if want_ray_sh:
        return enc_in, xyzs, ray_sh
    return enc_in, xyzs


def ray_sh_rows(rays_d):
    """[N,3] directions -> [N,16] half: each ray's degree-4 SH values as they stand in the colour-net input."""
    rays_d = rays_d.contiguous().float()
    N, dev = rays_d.shape[0], rays_d.device
    zeros, ones = torch.zeros(N, device=dev), torch.ones(N, device=dev)
    aabb = torch.tensor([-1., -1., -1., 1., 1., 1.], device=dev)
    return fixed_sample(torch.zeros_like(rays_d), rays_d, zeros, ones, aabb, None, 2, 1.0, want_ray_sh=True)[2]


class _density_head(Function):
    """h [M,16] half -> (weights [M], weights_sum [N], depth [N], sigma [M], cin [M,32] half)."""

    @staticmethod
    def forward(ctx, h, rays_d, nears, fars, noise, N, T, density_scale, obj_feat=None):
        """obj_feat: None -> cin [M,32]; [16] half (FOC network's encoded object feature) -> cin [M,48]."""
        h = h.contiguous()
        assert h.dtype == torch.float16 and h.shape == (N * T, 16)
        dev = h.device
        M = N * T
        sigma = torch.empty(M, dtype=torch.float32, device=dev)
        trans = torch.empty(M, dtype=torch.float32, device=dev)
        weights = torch.empty(M, dtype=torch.float32, device=dev)
        ws = torch.empty(N, dtype=torch.float32, device=dev)
        depth = torch.empty(N, dtype=torch.float32, device=dev)
        width = 32 if obj_feat is None else 48
        if obj_feat is not None:
            obj_feat = obj_feat.detach().reshape(-1).half().contiguous()
            assert obj_feat.numel() == 16
        cin = torch.empty(M, width, dtype=torch.float16, device=dev)
        check(lib.foc_fixed_head_forward(ptr(h), ptr(rays_d), ptr(nears), ptr(fars), ptr(noise), N, T, float(density_scale), ptr(sigma), ptr(trans),
                                         ptr(weights), ptr(ws), ptr(depth), ptr(cin), ptr(obj_feat), width, stream_of(h)), "fixed_head_forward")
        ctx.save_for_backward(h, sigma, trans, nears, fars, noise if noise is not None else torch.empty(0, device=dev))
        ctx.has_noise = noise is not None
        ctx.dims = (N, T, float(density_scale))
        ctx.width = width
        ctx.mark_non_differentiable(sigma)
        ctx.set_materialize_grads(False)
        return weights, ws, depth, sigma, cin

    @staticmethod
    def backward(ctx, g_weights, g_ws, g_depth, g_sigma, g_cin):
        h, sigma, trans, nears, fars, noise = ctx.saved_tensors
        N, T, ds = ctx.dims
        noise = noise if ctx.has_noise else None
        g_weights = g_weights.contiguous().float() if g_weights is not None else None
        g_ws = g_ws.contiguous().float() if g_ws is not None else None
        g_depth = g_depth.contiguous().float() if g_depth is not None else None
        g_cin = g_cin.contiguous().half() if g_cin is not None else None
        grad_h = torch.empty_like(h)
        check(lib.foc_fixed_head_backward(ptr(h), ptr(sigma), ptr(trans), ptr(nears), ptr(fars), ptr(noise), ptr(g_weights), ptr(g_ws), ptr(g_depth),
                                          ptr(g_cin), N, T, ds, ptr(grad_h), ctx.width, stream_of(h)), "fixed_head_backward")
        g_obj = None
        if ctx.width == 48 and ctx.needs_input_grad[8] and g_cin is not None:
            g_obj = g_cin[:, 31:47].float().sum(0)                 # one feature vector feeds every sample
        return grad_h, None, None, None, None, None, None, None, g_obj


class _fixed_composite(Function):
    """c [M,16] half (colour-net output), weights [M] -> image [N,3] = sum w*sigmoid(c)*[w>thresh] + (1 - sum w)*bg."""

    @staticmethod
    def forward(ctx, c, weights, bg_ray, bg_scalar, N, T, thresh):
        c = c.contiguous()
        weights = weights.contiguous()
        assert c.dtype == torch.float16 and c.shape == (N * T, 16) and weights.dtype == torch.float32
        image = torch.empty(N, 3, dtype=torch.float32, device=c.device)
        check(lib.foc_fixed_composite_forward(ptr(c), ptr(weights), ptr(bg_ray), float(bg_scalar), N, T, float(thresh), ptr(image), stream_of(c)),
              "fixed_composite_forward")
        ctx.save_for_backward(c, weights, bg_ray if bg_ray is not None else torch.empty(0, device=c.device))
        ctx.has_bg = bg_ray is not None
        ctx.dims = (N, T, float(thresh), float(bg_scalar))
        return image

    @staticmethod
    def backward(ctx, g_image):
        c, weights, bg_ray = ctx.saved_tensors
        N, T, thresh, bg_scalar = ctx.dims
        bg_ray = bg_ray if ctx.has_bg else None
        g_image = g_image.contiguous().float()
        grad_c = torch.empty_like(c)
        grad_w = torch.empty_like(weights)
        check(lib.foc_fixed_composite_backward(ptr(g_image), ptr(c), ptr(weights), ptr(bg_ray), bg_scalar, N, T, thresh, ptr(grad_c), ptr(grad_w),
                                               stream_of(c)), "fixed_composite_backward")
        return grad_c, grad_w, None, None, None, None, None


def tail_fusable(model):
    """Shapes `_render_tail` serves: 16-wide sigma head, degree-4 SH, 64-wide colour network of 2 or 3 layers, its input the 32-wide
    [SH16 | geo15 | 0] of network_ff.py or FOC's 48-wide [SH16 | geo15 | object feature 16 | 0] (network_tcnn.py:611-640)."""
    import os
    from .ffmlp import FFMLP
    from .shencoder import SHEncoder
    cn = getattr(model, "color_net", None)
    want_in = 48 if getattr(model, "uses_object_feature", False) else 32
    return (isinstance(cn, FFMLP) and cn.input_dim == want_in and cn.hidden_dim == 64 and cn.num_layers in (2, 3) and cn.padded_output_dim == 16
            and cn.activation in (0, 6)
            and isinstance(getattr(model, "encoder_dir", None), SHEncoder) and getattr(model, "geo_feat_dim", 0) == 15
            and (want_in == 32 or getattr(model, "yolo_encoding_dim", 0) == 16) and os.environ.get("FOC_FUSED_TAIL", "1") != "0")


_C_WIDTH = 4        # columns of the colour network's output that exist in memory on the fused tail (rgb logits + one pad)


class _render_tail(Function):
    """Density head -> colour network -> composite as ONE node: h [M,16] half + the colour network's weight blob ->
    image [N,3], weights_sum [N], depth [N] (+ sigma [M], weights [M], c [M,4] = rgb logits + one pad column, not differentiable).

    Same values and gradients as `_density_head` -> `FFMLP.forward_padded` -> `_fixed_composite`, bit for bit, without the colour
    network's input: its kernels read h and one SH row per ray (foc_color_head_forward), and in the backward pass the colour
    network writes grad_h itself — its input gradient for columns 1..15 merged with the density path's column 0 — so neither
    cin [M,32] nor grad_cin [M,32] exists (0.5 GB of traffic per 2 M-sample step). Head and composite are one kernel per
    direction (foc_fixed_tail_forward / _backward). ray_sh [N,16] half: `fixed_sample(..., want_ray_sh=True)` or `ray_sh_rows`.

    obj_feat [16] (or None): FOC's encoded object feature — the colour network then has 48-wide W0 rows; the feature's share of layer 0
    is a constant per neuron inside the kernels, its gradient (for the object-feature encoder) comes back as one [16] vector.
    want_sumsq: a seventh, differentiable output sumsq [N] = sum_t sigma^2 per ray (the samples' share of the outside-mask criterion)."""

    @staticmethod
    def forward(ctx, h, cweights, ray_sh, nears, fars, noise, bg_ray, bg_scalar, N, T, density_scale, thresh, num_layers, activation, obj_feat=None,
                want_sumsq=False, c_pre=None, w16_pre=None):
        # c_pre [M,4] half: the colour logits already computed from this h, these weights and this ray_sh by the encoder -> sigma node's fused
        # forward (field._hashgrid_mlp with `colour`, foc_field_forward_train: the bits foc_color_head_forward would give) — then no launch here
        from .field import _half_of
        h = h.contiguous()
        assert h.dtype == torch.float16 and h.shape == (N * T, 16)
        assert ray_sh.dtype == torch.float16 and ray_sh.shape == (N, 16) and ray_sh.is_contiguous()
        dev, M = h.device, N * T
        st = stream_of(h)
        w16 = w16_pre if w16_pre is not None else _half_of(cweights)     # w16_pre: the half copy the fused forward already made of THESE weights
        # of the colour network's 16 padded outputs only the rgb logits are ever read: they travel as [M,4] rows (as does their gradient)
        obj16 = None
        if obj_feat is not None:
            obj16 = obj_feat.detach().reshape(-1).half().contiguous()
            assert obj16.numel() == 16 and w16.numel() == 64 * (48 + 64 * (int(num_layers) - 1) + 16)
        if c_pre is not None:
            assert c_pre.dtype == torch.float16 and c_pre.shape == (M, _C_WIDTH) and c_pre.is_contiguous()
            c = c_pre
        else:
            c = torch.empty(M, _C_WIDTH, dtype=torch.float16, device=dev)
            check(lib.foc_color_head_forward(ptr(h), ptr(ray_sh), T, ptr(w16), M, 64, int(num_layers), int(activation), ptr(c), _C_WIDTH, ptr(obj16), st),
                  "color_head_forward")
        sigma = torch.empty(M, dtype=torch.float32, device=dev)
        trans = torch.empty(M, dtype=torch.float32, device=dev)
        weights = torch.empty(M, dtype=torch.float32, device=dev)
        ws = torch.empty(N, dtype=torch.float32, device=dev)
        depth = torch.empty(N, dtype=torch.float32, device=dev)
        image = torch.empty(N, 3, dtype=torch.float32, device=dev)
        sumsq = torch.empty(N, dtype=torch.float32, device=dev) if want_sumsq else None
        check(lib.foc_fixed_tail_forward(ptr(h), ptr(c), ptr(nears), ptr(fars), ptr(noise), ptr(bg_ray), float(bg_scalar), N, T, float(density_scale),
                                         float(thresh), ptr(sigma), ptr(trans), ptr(weights), ptr(ws), ptr(depth), ptr(image), _C_WIDTH, ptr(sumsq), st),
              "fixed_tail_forward")
        empty = torch.empty(0, device=dev)
        ctx.save_for_backward(h, w16, sigma, trans, weights, c, ray_sh, nears, fars, noise if noise is not None else empty,
                              bg_ray if bg_ray is not None else empty, obj16 if obj16 is not None else empty)
        ctx.flags = (noise is not None, bg_ray is not None, obj16 is not None, obj_feat.dtype if obj_feat is not None else None,
                     tuple(obj_feat.shape) if obj_feat is not None else None)
        ctx.dims = (N, T, float(density_scale), float(thresh), float(bg_scalar), int(num_layers), int(activation))
        ctx.mark_non_differentiable(sigma, weights, c)
        ctx.set_materialize_grads(False)          # unused outputs arrive as None in backward, not as five freshly zero-filled tensors (25 us)
        if want_sumsq:
            return image, ws, depth, sigma, weights, c, sumsq
        return image, ws, depth, sigma, weights, c

    @staticmethod
    def backward(ctx, g_image, g_ws, g_depth, _g_sigma, _g_weights, _g_c, g_sumsq=None):
        from .backend import _scratch
        h, w16, sigma, trans, weights, c, ray_sh, nears, fars, noise, bg_ray, obj16 = ctx.saved_tensors
        has_noise, has_bg, has_obj, obj_dtype, obj_shape = ctx.flags
        obj16 = obj16 if has_obj else None
        N, T, ds, thresh, bg_scalar, num_layers, activation = ctx.dims
        noise = noise if has_noise else None
        bg_ray = bg_ray if has_bg else None
        dev, M = h.device, N * T
        st = stream_of(h)
        g_image = g_image.contiguous().float() if g_image is not None else torch.zeros(N, 3, dtype=torch.float32, device=dev)
        g_ws = g_ws.contiguous().float() if g_ws is not None else None
        g_depth = g_depth.contiguous().float() if g_depth is not None else None
        g_sumsq = g_sumsq.contiguous().float() if g_sumsq is not None else None
        grad_c = torch.empty_like(c)
        grad_h0 = torch.empty(M, dtype=torch.float16, device=dev)
        check(lib.foc_fixed_tail_backward(ptr(g_image), ptr(g_ws), ptr(g_depth), ptr(c), ptr(sigma), ptr(trans), ptr(weights), ptr(nears), ptr(fars),
                                          ptr(noise), ptr(bg_ray), bg_scalar, N, T, ds, thresh, ptr(grad_c), ptr(grad_h0), _C_WIDTH, ptr(g_sumsq), st),
              "fixed_tail_backward")
        grad_h = torch.empty_like(h)
        g_w = torch.empty_like(w16)
        wsb = _scratch.get("ffmlp_ws", lib.foc_ffmlp_backward_workspace_bytes(48 if has_obj else 32, 64, num_layers), dev)
        g_obj32 = torch.empty(16, dtype=torch.float32, device=dev) if has_obj and ctx.needs_input_grad[14] else None
        check(lib.foc_color_head_backward(ptr(grad_c), ptr(h), ptr(ray_sh), T, ptr(grad_h0), ptr(w16), M, 64, num_layers, activation, ptr(grad_h),
                                          ptr(g_w), ptr(wsb), wsb.numel(), _C_WIDTH, ptr(obj16), ptr(g_obj32), st), "color_head_backward")
        g_obj = g_obj32.to(obj_dtype).view(obj_shape) if g_obj32 is not None else None
        return (grad_h, g_w) + (None,) * 12 + (g_obj, None, None, None)


class _masked_norm(Function):
    """sqrt(sum(sumsq[outside])) with the subgradient 0 at 0 that torch.norm has: FOC's `criterion_outside_mask`
    (nerf/renderer.py:163-165: torch.norm(sigma[~mask_rays])) from the per-ray sums of sigma^2 the tail kernel returns."""

    @staticmethod
    def forward(ctx, sumsq, outside):
        crit = torch.sqrt((sumsq * outside).sum())
        ctx.save_for_backward(crit, outside)
        return crit

    @staticmethod
    def backward(ctx, g):
        crit, outside = ctx.saved_tensors
        coef = torch.where(crit > 0, g / (2 * crit), torch.zeros_like(crit))
        return outside * coef, None


def _background(bg_color, N, dev):
    """bg_color None / scalar / tensor -> (per-ray [N,3] tensor or None, scalar)."""
    if bg_color is None:
        return None, 1.0
    if torch.is_tensor(bg_color):
        if bg_color.numel() > 1:
            return bg_color.to(dev, torch.float32).expand(N, 3).contiguous(), 0.0
        return None, float(bg_color)
    return None, float(bg_color)


def render_fixed_steps(model, rays_o, rays_d, yolo_details=None, num_steps=512, bg_color=None, perturb=False, weight_thresh=1e-10,
                       return_fields=None, _out=None, **kwargs):
    """Drop-in for NeRFRenderer.run(..., upsample_steps=0) on a focnerf_amd NeRFNetwork (fp16 autocast semantics); same result
    dictionary (`return_fields` None = on in eval mode, like the reference's run(), off in training)."""
    import time
    if return_fields is None:
        return_fields = not model.training
    t_start = time.time()
    prefix = rays_o.shape[:-1]
    rays_o = rays_o.contiguous().view(-1, 3).float()
    rays_d = rays_d.contiguous().view(-1, 3).float()
    N, T = rays_o.shape[0], int(num_steps)
    dev = rays_o.device
    aabb = model.aabb_train if model.training else model.aabb_infer
    nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, aabb, model.min_near)
    noise = torch.rand(N * T, dtype=torch.float32, device=dev) if perturb else None
    want_tail = (tail_fusable(model) and model.training and torch.is_grad_enabled())
    from .field import infer_fusable, field_infer
    fused_infer = not torch.is_grad_enabled() and infer_fusable(model)
    rb = ray_block_default() if fused_infer else 0
    enc_in, _, ray_sh = fixed_sample(rays_o, rays_d, nears, fars, aabb, noise, T, model.bound, want_ray_sh=True) if want_tail else \
        fixed_sample(rays_o, rays_d, nears, fars, aabb, noise, T, model.bound, ray_block=rb) + (None,)

    if fused_infer:
        # inference: sample -> encoder planes -> whole-field kernel -> weights + mask + composite kernel; between the kernels the samples
        # stand in 64-ray blocks (neighbouring rays at one depth on the lanes of a wave: the encoder's gathers share cache lines)
        obj_feat = model.encode_object_feature(yolo_details, dev) if getattr(model, "uses_object_feature", False) else None
        sigma, rgb = field_infer(model, enc_in, rays_d, dir_div=T, dir_block=rb, obj_feat=obj_feat)
        bg_ray, bg_scalar = _background(bg_color, N, dev)
        # `_out` = (depth [N], image [N,3]) fp32 contiguous views of the caller's whole-view buffers (NeRFRenderer.render, staged)
        direct = (_out is not None and _out[0].dtype == torch.float32 and _out[1].dtype == torch.float32 and _out[0].is_contiguous()
                  and _out[1].is_contiguous() and _out[0].numel() == N and _out[1].numel() == 3 * N and _out[0].device == dev)
        depth, image = (_out[0].view(N), _out[1].view(N, 3)) if direct else (torch.empty(N, dtype=torch.float32, device=dev),
                                                                              torch.empty(N, 3, dtype=torch.float32, device=dev))
        weights_sum = torch.empty(N, dtype=torch.float32, device=dev)
        rgb_masked = densities = None
        if return_fields:
            # the whole-view field buffers of a staged render (`_out[2:]` = densities [N,T], rgbs [N,T,3] views): written in place
            if direct and len(_out) == 4 and all(t.dtype == torch.float32 and t.is_contiguous() and t.device == dev for t in _out[2:]) \
                    and _out[2].numel() == N * T and _out[3].numel() == 3 * N * T and rb:
                densities, rgb_masked = _out[2].view(N * T), _out[3].view(N * T, 3)
            else:
                rgb_masked = torch.empty(N * T, 3, dtype=torch.float32, device=dev)
                densities = torch.empty(N * T, dtype=torch.float32, device=dev) if rb else None
        check(lib.foc_fixed_render_inference(ptr(sigma), ptr(rgb), ptr(nears), ptr(fars), ptr(noise), ptr(bg_ray), float(bg_scalar), N, T,
                                             float(model.density_scale), float(weight_thresh), ptr(image), ptr(depth), ptr(weights_sum), ptr(rgb_masked),
                                             rb, ptr(densities), stream_of(sigma)), "fixed_render_inference")
        t_mid = time.time()
        results = {'depth': depth.view(*prefix), 'image': image.view(*prefix, 3), 'weights_sum': weights_sum, 'criterion_outside_mask': None,
                   'timing': [t_mid - t_start, time.time() - t_mid]}
        if return_fields:
            results['densities'] = (densities if rb else sigma).view(N, T, 1)
            results['rgbs'] = rgb_masked.view(N, T, 3)
        return results

    enc = model.encoder
    from .field import field_fusable, _hashgrid_mlp, colour_forward_fusable
    c_pre = obj_feat = None
    uses_obj = getattr(model, "uses_object_feature", False)
    with torch.autocast("cuda", dtype=torch.float16):
        if uses_obj:                                                      # FOC network (network_foc.py): encoded YOLO feature in the colour input
            obj_feat = model.encode_object_feature(yolo_details, dev)
        if field_fusable(enc, model.sigma_net):
            import numpy as np
            mlp = model.sigma_net
            # with the fused tail the colour network's forward rides in the sigma network's kernel (its logits reach _render_tail as `c_pre`)
            colour = wc16 = None
            if want_tail and colour_forward_fusable(mlp, model.color_net, uses_obj):
                from .field import _half_of
                wc16 = _half_of(model.color_net.weights)                  # ONE half copy per step for both nodes that read the colour weights
                colour = (wc16, ray_sh, T, model.color_net.num_layers, _C_WIDTH, obj_feat)
            h = _hashgrid_mlp.apply(enc_in, enc.embeddings, mlp.weights, enc.offsets,
                                    (float(np.log2(enc.per_level_scale)), enc.base_resolution, enc.gridtype_id, enc.align_corners, enc.interp_id),
                                    (mlp.input_dim, mlp.hidden_dim, mlp.num_layers, mlp.activation, mlp.output_activation),
                                    mlp.training and torch.is_grad_enabled(), colour)
            if colour is not None:
                h, c_pre = h
        else:
            feats = grid_encode(enc_in, enc.embeddings, enc.offsets, enc.per_level_scale, enc.base_resolution, False, enc.gridtype_id,
                                enc.align_corners, enc.interp_id)
            h = model.sigma_net(feats)                                    # [M,16] half
        if h.shape[1] != 16:                                              # FFMLP slices to output_dim (= 16 here: 1 + geo_feat_dim 15)
            raise RuntimeError("render_fixed_steps expects a 16-wide sigma head (1 + geo_feat_dim = 16)")
        fused_tail = want_tail
        # nerf/renderer.py:163-165: in training with an object mask, the norm of the densities of the rays outside it
        want_crit = model.training and yolo_details is not None and yolo_details[0] is not None
        criterion_outside_mask = None
        if fused_tail:
            cn = model.color_net
            bg_ray, bg_scalar = _background(bg_color, N, dev)
            outs = _render_tail.apply(h, cn.weights, ray_sh, nears, fars, noise, bg_ray, bg_scalar, N, T, model.density_scale, weight_thresh,
                                      cn.num_layers, cn.activation, obj_feat, want_crit and yolo_details[0].numel() == N, c_pre,
                                      wc16 if c_pre is not None else None)
            image, weights_sum, depth, sigma, weights, c = outs[:6]
            if len(outs) == 7:
                # a per-ray mask: the samples' sum of sigma^2 comes out of the tail kernel (no [M]-sized torch expression, no boolean-mask
                # indexing with its host sync); sigma here is the kernel's exp(h0) — density_scale == 1 in every FOC configuration, and the
                # criterion is taken on trunc_exp(h0) itself like the reference's
                criterion_outside_mask = _masked_norm.apply(outs[6], (~yolo_details[0].reshape(N)).to(torch.float32))
        else:
            weights, weights_sum, depth, sigma, cin = _density_head.apply(h, rays_d, nears, fars, noise, N, T, model.density_scale, obj_feat)
        if want_crit and criterion_outside_mask is None:
            from .activation import trunc_exp
            criterion_outside_mask = torch.norm(trunc_exp(h[:, 0]).view(N, T)[~yolo_details[0].squeeze(0)] - 0)
        if not fused_tail:
            c = model.color_net.forward_padded(cin)                        # [M,16] half, columns 0..2 = rgb logits
    t_mid = time.time()

    if not fused_tail:
        bg_ray, bg_scalar = _background(bg_color, N, dev)
        image = _fixed_composite.apply(c, weights, bg_ray, bg_scalar, N, T, weight_thresh)

    results = {'depth': depth.view(*prefix), 'image': image.view(*prefix, 3), 'weights_sum': weights_sum,
               'criterion_outside_mask': criterion_outside_mask, 'timing': [t_mid - t_start, time.time() - t_mid]}
    if return_fields:
        from .head import rgb_head
        c16 = c.detach()
        if c16.shape[1] != 16:                                                    # fused tail: [M,4] logits -> the 16-wide rows rgb_head reads
            c16 = torch.nn.functional.pad(c16, (0, 16 - c16.shape[1]))
        rgb = rgb_head(c16) * (weights > weight_thresh).unsqueeze(-1)            # half-rounded sigmoid, fp32 tensor (the composite's values)
        results['densities'] = sigma.view(N, T, 1)
        results['rgbs'] = rgb.view(N, T, 3)
    return results


@torch.no_grad()
def render_field4(model, rays_o, rays_d, num_steps=512, weight_thresh=1e-10, yolo_details=None, out=None):
    """What the combiner needs from one object for a chunk of rays — COMBINED.py's `run` (:451-534 with upsample_steps=0, perturb off):
    `densities` [N,T] and `rgbs` [N,T,3] (zero where the object's own compositing weight is <= 1e-10) — PACKED as field4 [N,T,4] fp32
    (sigma, r, g, b), written into `out` when given. Fused path: sample -> encoder -> whole-field kernel -> weights + mask + pack
    (foc_fixed_field_pack); other networks go through `model.run(..., return_fields=True)` and are packed with torch ops."""
    from .field import infer_fusable, field_infer
    rays_o = rays_o.contiguous().view(-1, 3).float()
    rays_d = rays_d.contiguous().view(-1, 3).float()
    N, T = rays_o.shape[0], int(num_steps)
    dev = rays_o.device
    if out is None:
        out = torch.empty(N, T, 4, dtype=torch.float32, device=dev)
    assert out.shape == (N, T, 4) and out.dtype == torch.float32 and out.is_contiguous()
    if infer_fusable(model) and model.bg_radius <= 0:
        aabb = model.aabb_train if model.training else model.aabb_infer
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, aabb, model.min_near)
        rb = ray_block_default()
        enc_in, _ = fixed_sample(rays_o, rays_d, nears, fars, aabb, None, T, model.bound, ray_block=rb)
        obj_feat = model.encode_object_feature(yolo_details, dev) if getattr(model, "uses_object_feature", False) else None
        sigma, rgb = field_infer(model, enc_in, rays_d, dir_div=T, dir_block=rb, obj_feat=obj_feat)
        check(lib.foc_fixed_field_pack(ptr(sigma), ptr(rgb), ptr(nears), ptr(fars), None, None, 1.0, N, T, float(model.density_scale),
                                       float(weight_thresh), None, None, None, ptr(out), rb, stream_of(sigma)), "fixed_field_pack")
        return out
    from .combine import pack_field4
    with torch.autocast("cuda", dtype=torch.float16):
        res = model.run(rays_o[None], rays_d[None], yolo_details, num_steps=T, upsample_steps=0, perturb=False, weight_thresh=weight_thresh,
                        return_fields=True)
    out.copy_(pack_field4(res['densities'], res['rgbs']))
    return out
