"""Fused occupancy-grid TRAINING path: what `NeRFRenderer.run_cuda` computes in training mode (legacy/nerf/renderer.py:256-322) for a
`focnerf_amd.network.NeRFNetwork`, as ONE autograd node over the kernels

    march_rays_train (field layout) -> grid_encode (+ the backward's count pass) -> sigma_net -> color_net (head form) -> tail

instead of the caller-side chain  march_rays_train -> (x + bound) / (2 bound) -> grid_encode -> sigma_net -> trunc_exp / SH / cat / pad
-> color_net -> sigmoid -> composite_rays_train -> background / depth normalisation  with an autograd node, a handful of torch kernels
and their Python per stage. Same sample list (bit for bit), same values as that chain (the colour network's input is never materialised:
its first k-chunk is one SH row per sample written by the march's emit pass, its second the density network's output row, csrc/ffmlp.hip
MlpHead; sigma, rgb and their gradients stay on the lane in csrc/occtrain.hip), same gradients up to fp32 summation order.

The step it replaces was bound by its launches: ~85 kernels, 0.9 ms of GPU time, 1.2 ms of host time to enqueue them from Python.
FOC_FUSED_OCC=0 keeps the chain (tests compare the two).
"""
import os

import numpy as np
import torch
from torch.autograd import Function

import ctypes

from ._lib import lib, ptr, stream_of, check, FocOccTrainNode, FOC_F16
from .backend import _gridencoder, _ffmlp, _scratch

_C_WIDTH = 4


def occ_train_fusable(model):
    """Shapes the node serves: hash grid (D 3, C 2) -> FFMLP density network with a 16-wide output -> degree-4 SH + 15 geometry features
    -> 64-wide FFMLP colour network of 2 or 3 layers, no background model, no object feature."""
    from .field import field_fusable
    from .fixedstep import tail_fusable
    return (field_fusable(model.encoder, model.sigma_net) and tail_fusable(model) and not getattr(model, "uses_object_feature", False)
            and model.bg_radius <= 0 and model.sigma_net.activation == model.color_net.activation
            and os.environ.get("FOC_FUSED_OCC", "1") != "0")


def _round_up(count, align):
    return count + (align - count % align) if align > 0 else count            # raymarching.py:190,226


_zero_jitter = {}


def _no_jitter(n, dev):
    """A [n] block of zeros per device, made once (`perturb` off: the march adds 0 * dt to every ray's start)."""
    z = _zero_jitter.get(dev)
    if z is None or z.shape[0] < n:
        z = _zero_jitter[dev] = torch.zeros(max(n, 4096), dtype=torch.float32, device=dev)
    return z[:n]


_plan_bytes = {}


def _native_plan(offsets, S, H, L, gridtype, M, sig_cfg, col_cfg):
    """(grid workspace bytes, MLP workspace bytes) when the node can run as ONE library call each way (include/focnerf.h FocOccTrainNode:
    the encoder's counted forward and binned backward must apply, the switches that take other paths must be at their defaults), else None.
    FOC_OCC_NATIVE_NODE=0: always the call-by-call chain below (the tests compare the two)."""
    if (os.environ.get("FOC_OCC_NATIVE_NODE", "1") == "0" or os.environ.get("FOC_GRID_PRECOUNT", "1") != "1"
            or os.environ.get("FOCNERF_GRID_ATOMIC", "0") == "1"):
        return None
    key = (M, L, sig_cfg[:3], col_cfg[0])
    sizes = _plan_bytes.get(key)
    if sizes is None:
        sizes = _plan_bytes[key] = (int(lib.foc_grid_encode_backward_workspace_bytes(M, 3, 2, L, FOC_F16)),
                                    max(int(lib.foc_ffmlp_backward_workspace_bytes(32, 64, int(col_cfg[0]))),
                                        int(lib.foc_ffmlp_backward_workspace_bytes(int(sig_cfg[0]), int(sig_cfg[1]), int(sig_cfg[2])))))
    if not sizes[0] or M * 8 * L >= 2 ** 32 or not _gridencoder._binned_ok(offsets, S, H, L, gridtype):
        return None
    return sizes


def _a(t):
    return t.data_ptr() if t is not None else None


class _occ_train(Function):
    @staticmethod
    def forward(ctx, emb, w_sigma, w_color, o, d, aabb, bitfield, counter, bg_ray, cfg):
        from .field import _half_of
        (bound, cascade, grid_size, mean_count, perturb, align, force_all_rays, dt_gamma, max_steps, T_thresh, density_scale, bg_scalar,
         offsets, enc_cfg, sig_cfg, col_cfg, min_near) = cfg
        S, H, gridtype, align_corners, interp = enc_cfg
        n, dev = o.shape[0], o.device
        st = stream_of(o)
        budgeted = mean_count > 0 and not force_all_rays
        cap = _round_up(mean_count, align) if budgeted else n * max_steps
        # one block for the three sample arrays: enc_in [cap,3] fp32 | deltas [cap,2] fp32 | sh [cap,16] fp16 (every row written by the emit pass)
        o1, o2 = -(-3 * cap // 4) * 4, -(-3 * cap // 4) * 4 + -(-2 * cap // 4) * 4         # sections start on 16-byte boundaries
        block = torch.empty(o2 + 8 * cap, dtype=torch.float32, device=dev)
        enc_in, deltas, sh = block[: 3 * cap].view(cap, 3), block[o1: o1 + 2 * cap].view(cap, 2), block[o2:].view(torch.float16).view(cap, 16)
        rays = torch.empty(n, 3, dtype=torch.int32, device=dev)
        nf = torch.empty(2, n, dtype=torch.float32, device=dev)         # nears, fars: written by the march's count pass (the box test of near_far_from_aabb)
        nears, fars = nf[0], nf[1]
        jitter = torch.rand(n, dtype=torch.float32, device=dev) if perturb else _no_jitter(n, dev)
        scratch = _scratch.get("march", lib.foc_march_rays_train_scratch_bytes(n, max_steps), dev)
        L = offsets.shape[0] - 1
        plan = _native_plan(offsets, S, H, L, gridtype, cap, sig_cfg, col_cfg) if (budgeted and cap > 0 and n > 0) else None
        if plan is not None:
            # the whole forward as one library call (csrc/occtrain.hip foc_occ_train_forward): the same five entry points in the same order,
            # enqueued from C — the step's host time no longer depends on nine trips through the binding
            M = cap
            emb16, ws16, wc16 = _half_of(emb), _half_of(w_sigma), _half_of(w_color)
            planes = torch.empty(L, M, 2, dtype=torch.float16, device=dev)
            hc = torch.empty(M * (16 + _C_WIDTH), dtype=torch.float16, device=dev)
            h, c = hc[: M * 16].view(M, 16), hc[M * 16:].view(M, _C_WIDTH)
            out = torch.empty(n * 8, dtype=torch.float32, device=dev)
            ws, depth, image_raw, image = out[:n], out[n: 2 * n], out[2 * n: 5 * n].view(n, 3), out[5 * n:].view(n, 3)
            gws = _scratch.get("grid_bwd", plan[0], dev)
            nd = FocOccTrainNode()
            nd.struct_bytes = ctypes.sizeof(FocOccTrainNode)
            nd.n_rays, nd.max_steps, nd.cascade, nd.grid_size, nd.cap, nd.pad_align = n, int(max_steps), int(cascade), int(grid_size), M, 0
            nd.bound, nd.dt_gamma, nd.min_near = float(bound), float(dt_gamma), float(min_near)
            nd.rays_o, nd.rays_d, nd.aabb, nd.jitter, nd.bitfield = _a(o), _a(d), _a(aabb), _a(jitter), _a(bitfield)
            nd.nears, nd.fars, nd.enc_in, nd.deltas, nd.sh_rows = _a(nears), _a(fars), _a(enc_in), _a(deltas), _a(sh)
            nd.rays, nd.counter, nd.march_scratch = _a(rays), _a(counter), _a(scratch)
            nd.levels, nd.base_resolution, nd.gridtype, nd.interp = L, int(H), int(gridtype), int(interp)
            nd.align_corners, nd.table_dtype, nd.per_level_scale_log2 = int(bool(align_corners)), FOC_F16, float(S)
            nd.embeddings, nd.offsets, nd.offsets_host = _a(emb16), _a(offsets), _gridencoder._host_offsets(offsets)
            nd.planes, nd.grid_workspace, nd.grid_workspace_bytes = _a(planes), _a(gws), plan[0]
            nd.sigma_input_dim, nd.sigma_hidden, nd.sigma_layers, nd.sigma_activation, nd.sigma_output_activation = (
                int(sig_cfg[0]), int(sig_cfg[1]), int(sig_cfg[2]), int(sig_cfg[3]), 6)
            nd.color_hidden, nd.color_layers, nd.color_activation, nd.c_width = 64, int(col_cfg[0]), int(col_cfg[1]), _C_WIDTH
            nd.w_sigma, nd.w_color, nd.h, nd.c = _a(ws16), _a(wc16), _a(h), _a(c)
            nd.T_thresh, nd.density_scale, nd.bg_scalar, nd.bg_ray = float(T_thresh), float(density_scale), float(bg_scalar), _a(bg_ray)
            nd.weights_sum, nd.image_raw, nd.image, nd.depth = _a(ws), _a(image_raw), _a(image), _a(depth)
            check(lib.foc_occ_train_forward(ctypes.byref(nd), st), "occ_train_forward")
            # the count pass rode in the encoder's forward: the ticket the backward checks, as backend.grid_encode_forward_counted issues it
            idx, pre = _gridencoder._pre_state(dev)
            pre["ticket"] += 1
            pre["key"] = (enc_in.data_ptr(), M, L, FOC_F16, gws.data_ptr())
            ctx.save_for_backward(enc_in, emb16, ws16, wc16, offsets, planes, h, c, sh, deltas, rays, counter, ws, image_raw,
                                  bg_ray if bg_ray is not None else torch.empty(0, device=dev))
            ctx.nears_fars = nf
            ctx.cfg = (M, n, float(T_thresh), float(density_scale), float(bg_scalar), bg_ray is not None, enc_cfg, sig_cfg, col_cfg)
            ctx.ticket = (idx, pre["ticket"], pre["key"])
            ctx.node, ctx.plan = nd, plan
            ctx.mark_non_differentiable(depth)
            ctx.set_materialize_grads(False)
            return image, ws, depth
        ctx.node = None
        check(lib.foc_march_rays_train_field(ptr(o), ptr(d), ptr(bitfield), float(bound), float(dt_gamma), int(max_steps), n, int(cascade), int(grid_size), cap,
                                             ptr(nears), ptr(fars), ptr(enc_in), ptr(sh), ptr(deltas), ptr(rays), ptr(counter), ptr(jitter), ptr(scratch),
                                             0 if budgeted else max(int(align), 1), ptr(aabb), float(min_near), st), "march_rays_train_field")
        M = cap
        if not budgeted:                                        # raymarching.py:223-229: the list is cut to the samples marched (one device -> host copy)
            M = min(cap, _round_up(int(counter[0].item()), align))
            enc_in, deltas, sh = enc_in[:M], deltas[:M], sh[:M]
        L = offsets.shape[0] - 1
        emb16, ws16, wc16 = _half_of(emb), _half_of(w_sigma), _half_of(w_color)
        planes = torch.empty(L, M, 2, dtype=torch.float16, device=dev)
        ticket = _gridencoder.grid_encode_forward_counted(enc_in, emb16, offsets, planes, M, 3, 2, L, S, H, gridtype, align_corners, interp) if M else None
        if ticket is None and M:
            _gridencoder.grid_encode_forward(enc_in, emb16, offsets, planes, M, 3, 2, L, S, H, None, gridtype, align_corners, interp)
        h = torch.empty(M, 16, dtype=torch.float16, device=dev)
        if M:
            _ffmlp.ffmlp_forward_planar(planes, ws16, M, sig_cfg[0], 16, sig_cfg[1], sig_cfg[2], sig_cfg[3], 6, h)
        c = torch.empty(M, _C_WIDTH, dtype=torch.float16, device=dev)
        if M:
            check(lib.foc_color_head_forward(ptr(h), ptr(sh), 1, ptr(wc16), M, 64, int(col_cfg[0]), int(col_cfg[1]), ptr(c), _C_WIDTH, None, st), "color_head_forward")
        out = torch.empty(n * 8, dtype=torch.float32, device=dev)
        ws, depth, image_raw, image = out[:n], out[n: 2 * n], out[2 * n: 5 * n].view(n, 3), out[5 * n:].view(n, 3)
        check(lib.foc_occ_tail_forward(ptr(h), ptr(c), _C_WIDTH, ptr(deltas), ptr(rays), M, n, float(T_thresh), float(density_scale), ptr(bg_ray), float(bg_scalar),
                                       ptr(nears), ptr(fars), ptr(ws), ptr(image_raw), ptr(image), ptr(depth), st), "occ_tail_forward")
        ctx.save_for_backward(enc_in, emb16, ws16, wc16, offsets, planes, h, c, sh, deltas, rays, counter, ws, image_raw,
                              bg_ray if bg_ray is not None else torch.empty(0, device=dev))
        ctx.nears_fars = nf
        ctx.cfg = (M, n, float(T_thresh), float(density_scale), float(bg_scalar), bg_ray is not None, enc_cfg, sig_cfg, col_cfg)
        ctx.ticket = ticket
        ctx.mark_non_differentiable(depth)
        ctx.set_materialize_grads(False)
        return image, ws, depth

    @staticmethod
    def backward(ctx, g_image, g_ws, _g_depth):
        enc_in, emb16, ws16, wc16, offsets, planes, h, c, sh, deltas, rays, counter, ws, image_raw, bg_ray = ctx.saved_tensors
        M, n, T_thresh, density_scale, bg_scalar, has_bg, enc_cfg, sig_cfg, col_cfg = ctx.cfg
        S, H, gridtype, align_corners, interp = enc_cfg
        dev = h.device
        st = stream_of(h)
        L = offsets.shape[0] - 1
        g_emb = torch.zeros_like(emb16)
        g_wsig, g_wcol = torch.empty_like(ws16), torch.empty_like(wc16)
        if M == 0 or (g_image is None and g_ws is None):
            return g_emb, g_wsig.zero_(), g_wcol.zero_(), None, None, None, None, None, None, None, None
        g_image = g_image.contiguous().float() if g_image is not None else torch.zeros(n, 3, dtype=torch.float32, device=dev)
        g_ws = g_ws.contiguous().float() if g_ws is not None else None
        nd = getattr(ctx, "node", None)
        if nd is not None:                                  # the whole backward as one library call (foc_occ_train_backward)
            gws = _scratch.get("grid_bwd", ctx.plan[0], dev)
            mws = _scratch.get("ffmlp_ws", ctx.plan[1], dev)
            gblock = torch.empty(M * (_C_WIDTH + 1), dtype=torch.float16, device=dev)  # grad_c [M,4] | grad_h0 [M]: every row written by the kernel
            grad_h = torch.empty_like(h)                    # its own block: 32-byte rows written with 16-byte stores, M need not be a multiple of 8
            g_planes = torch.empty_like(planes)
            nd.grad_image, nd.grad_ws = _a(g_image), _a(g_ws)
            nd.grad_c, nd.grad_h0, nd.grad_h = gblock.data_ptr(), gblock.data_ptr() + 2 * M * _C_WIDTH, _a(grad_h)
            nd.grad_planes, nd.grad_w_color, nd.grad_w_sigma, nd.grad_embeddings = _a(g_planes), _a(g_wcol), _a(g_wsig), _a(g_emb)
            nd.mlp_workspace, nd.mlp_workspace_bytes, nd.grid_workspace, nd.grid_workspace_bytes = _a(mws), mws.numel(), _a(gws), ctx.plan[0]
            nd.precounted = int(_gridencoder._precount_valid(ctx.ticket, enc_in, M, L, FOC_F16, gws))
            check(lib.foc_occ_train_backward(ctypes.byref(nd), st), "occ_train_backward")
            _gridencoder._invalidate_precount(dev)          # the header now belongs to this pass (and a used ticket is spent)
            ctx.node = None
            return g_emb, g_wsig, g_wcol, None, None, None, None, None, None, None, None
        gblock = torch.empty(M * (_C_WIDTH + 1), dtype=torch.float16, device=dev)      # grad_c [M,4] | grad_h0 [M]: every row written by the kernel
        grad_c, grad_h0 = gblock[: M * _C_WIDTH].view(M, _C_WIDTH), gblock[M * _C_WIDTH:]
        check(lib.foc_occ_tail_backward(ptr(g_image), ptr(g_ws), ptr(h), ptr(c), _C_WIDTH, ptr(deltas), ptr(rays), ptr(counter), ptr(ws), ptr(image_raw), M, n,
                                        T_thresh, density_scale, ptr(bg_ray if has_bg else None), bg_scalar, ptr(grad_c), ptr(grad_h0), st), "occ_tail_backward")
        grad_h = torch.empty_like(h)
        wsb = _scratch.get("ffmlp_ws", lib.foc_ffmlp_backward_workspace_bytes(32, 64, int(col_cfg[0])), dev)
        check(lib.foc_color_head_backward(ptr(grad_c), ptr(h), ptr(sh), 1, ptr(grad_h0), ptr(wc16), M, 64, int(col_cfg[0]), int(col_cfg[1]), ptr(grad_h), ptr(g_wcol),
                                          ptr(wsb), wsb.numel(), _C_WIDTH, None, None, st), "color_head_backward")
        g_planes = torch.empty_like(planes)
        _ffmlp.ffmlp_backward_planar(grad_h, planes, ws16, M, sig_cfg[0], 16, sig_cfg[1], sig_cfg[2], sig_cfg[3], 6, True, g_planes, g_wsig)
        _gridencoder.grid_encode_backward(g_planes, enc_in, emb16, offsets, g_emb, M, 3, 2, L, S, H, None, None, gridtype, align_corners, interp, grad_bl=False,
                                          precount=ctx.ticket)
        return g_emb, g_wsig, g_wcol, None, None, None, None, None, None, None, None


def render_occupancy_train(model, o, d, counter, bg_color, perturb, force_all_rays, dt_gamma, max_steps, T_thresh, align):
    """o, d [n,3] fp32 contiguous, counter int32[2] (zeroed by the caller) -> (image [n,3], weights_sum [n], depth [n]); the rays' box test
    against the model's training box (near_far_from_aabb, min_near) happens inside the march."""
    from .fixedstep import _background
    enc, sn, cn = model.encoder, model.sigma_net, model.color_net
    n, dev = o.shape[0], o.device
    bg_ray, bg_scalar = _background(bg_color, n, dev)
    cfg = (float(model.bound), int(model.cascade), int(model.grid_size), int(model.mean_count), bool(perturb), int(align), bool(force_all_rays), float(dt_gamma),
           int(max_steps), float(T_thresh), float(model.density_scale), float(bg_scalar), enc.offsets,
           (float(np.log2(enc.per_level_scale)), enc.base_resolution, enc.gridtype_id, enc.align_corners, enc.interp_id),
           (sn.input_dim, sn.hidden_dim, sn.num_layers, sn.activation), (cn.num_layers, cn.activation), float(model.min_near))
    return _occ_train.apply(enc.embeddings, sn.weights, cn.weights, o, d, model._aabb().contiguous().float(), model.density_bitfield, counter, bg_ray, cfg)
