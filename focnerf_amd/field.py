"""Hash-grid encoder -> fused MLP as ONE autograd node, with the encoding kept in the encoder's native layout.

The reference's encoder kernels work on [L, B, C] (gridencoder.cu:218, :283) and its Python wrapper permutes + copies to
[B, L*C] for the MLP and back for the gradient (grid.py:57, :75). `grid_encode` here avoids the copies by letting the
encoder kernels address [B, L*C] directly, but a level-major launch that keeps one level's table in one XCD's L2 wants to
write [L, B, C] planes (0.59 vs 0.74 ms per 2 M points on MI355X). This node keeps the planes and lets the MLP kernels read /
write them (`foc_ffmlp_forward_planar`, `foc_ffmlp_backward_planar`), so neither side pays for the other's layout:

    h = hashgrid_mlp(encoder, mlp, x)        # == mlp.forward_padded(encoder(x, bound))   (same values, same gradients)

Used by NeRFNetwork (fused head) and render_fixed_steps; FOC_FUSED_FIELD=0 restores the two separate nodes.
"""
import os

import numpy as np
import torch
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from .backend import _gridencoder, _ffmlp
from .ffmlp import FFMLP, _fused_backward_ok
from .gridencoder import GridEncoder


def field_fusable(encoder, mlp):
    return (isinstance(encoder, GridEncoder) and isinstance(mlp, FFMLP) and encoder.input_dim == 3 and encoder.level_dim == 2
            and mlp.input_dim == encoder.output_dim and _fused_backward_ok(mlp.input_dim, mlp.hidden_dim, mlp.num_layers, mlp.activation)
            and mlp.padded_output_dim == 16 and os.environ.get("FOC_FUSED_FIELD", "1") != "0")


def _raw_stream_of(device):
    """Handle of torch's current stream on `device` (an int: cheap to take and to compare)."""
    from ._lib import raw_stream
    return raw_stream(device.index if device.index is not None else torch.cuda.current_device())


_half_scope = None          # dict id(param) -> (param, fp16 copy) while a `half_cache_scope()` is open, else None


class half_cache_scope:
    """Within the scope the fp16 copies of parameters are made once and reused: a staged render evaluates the same 50 MB table and the
    same weight blobs for each of its 157 ray chunks, under `no_grad`, with nothing writing the parameters in between. The scope is the
    ONLY cache: outside it every forward converts again, as the reference does on every call (grid.py:41-44; ffmlp.py:23) — validity
    cannot be inferred from the parameter itself, because writes through `.data` (torch_ema's copy_to / restore around every
    evaluation, nerf/utils.py:1164-1174; `reset_parameters`) leave its version counter untouched. Re-entrant; the outermost exit drops
    the copies."""

    def __enter__(self):
        global _half_scope
        self._outer = _half_scope
        if _half_scope is None:
            _half_scope = {}
        return self

    def __exit__(self, *exc):
        global _half_scope
        _half_scope = self._outer
        return False


def _half_of(param):
    """fp16 copy of a parameter: a fresh conversion, unless a `half_cache_scope` is open (then one conversion per scope)."""
    if param.dtype == torch.half:
        return param.contiguous()
    if _half_scope is None:
        return param.detach().to(torch.half).contiguous()
    hit = _half_scope.get(id(param))
    if hit is not None and hit[0] is param and hit[1].device == param.device:
        if hit[2] is not None and _raw_stream_of(param.device) != hit[3]:
            torch.cuda.current_stream(param.device).wait_event(hit[2])      # made on another stream (staged render: chunks alternate streams)
        return hit[1]
    h = param.detach().to(torch.half).contiguous()
    ev = st = None
    if h.is_cuda:
        st = _raw_stream_of(h.device)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(h.device))         # the stream of the TENSOR's device: it need not be the current device
    _half_scope[id(param)] = (param, h, ev, st)
    return h


def scope_cached(key, owner, make):
    """`make()` -> tensor, once per open `half_cache_scope` for (key, owner) — `owner` is kept and compared by identity so that a recycled
    id() cannot alias. Like `_half_of`, a value made on one stream is handed to another stream behind an event. No scope: `make()`."""
    if _half_scope is None:
        return make()
    hit = _half_scope.get(key)
    if hit is not None and hit[0] is owner:
        if hit[2] is not None and _raw_stream_of(hit[1].device) != hit[3]:
            torch.cuda.current_stream(hit[1].device).wait_event(hit[2])
        return hit[1]
    v = make()
    ev = st = None
    if v.is_cuda:
        st = _raw_stream_of(v.device)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(v.device))
    _half_scope[key] = (owner, v, ev, st)
    return v


def colour_forward_fusable(sigma_net, color_net, obj=False):
    """Shapes `foc_field_forward_train` serves (csrc/field_fwd.hip): both networks' training forward in one kernel — bit for bit
    `foc_ffmlp_forward_planar` + `foc_color_head_forward`. FOC_FUSED_FWD=0: the two calls (parity aid)."""
    return (isinstance(sigma_net, FFMLP) and isinstance(color_net, FFMLP) and sigma_net.input_dim == 32 and sigma_net.hidden_dim == 64
            and color_net.hidden_dim == 64 and color_net.input_dim == (48 if obj else 32) and sigma_net.padded_output_dim == 16
            and (sigma_net.num_layers, color_net.num_layers) in ((2, 2), (2, 3), (3, 3)) and sigma_net.activation == color_net.activation
            and sigma_net.activation in (0, 6) and sigma_net.output_activation == 6 and os.environ.get("FOC_FUSED_FWD", "1") != "0")


class _hashgrid_mlp(Function):
    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, x, embeddings, weights, offsets, enc_cfg, mlp_cfg, training, colour=None):
        # x [B,3] fp32 in [0,1]; embeddings [rows,2]; weights: FFMLP blob
        # colour = (colour weights, ray_sh [B / T, 16] half, T, colour layers, c_width, obj_feat or None): the colour network's forward runs in the
        # SAME kernel as the sigma network's (foc_field_forward_train) and its logits come back as a second, non-differentiable output — the
        # node that owns the colour network (fixedstep._render_tail) takes them instead of launching foc_color_head_forward, and computes every
        # gradient of the colour network in its own backward as before
        S, H, gridtype, align_corners, interp = enc_cfg
        input_dim, hidden_dim, num_layers, activation, output_activation = mlp_cfg
        x = x.contiguous().float()
        B = x.shape[0]
        L = offsets.shape[0] - 1
        emb = _half_of(embeddings)                              # grid.py:41-44: half table under autocast (C even)
        w = _half_of(weights)                                   # ffmlp.py:23: custom_fwd(cast_inputs=half)
        enc = torch.empty(L, B, 2, device=x.device, dtype=torch.half)
        # training: the backward's count pass rides along in the forward launch (backend.grid_encode_forward_counted)
        ticket = _gridencoder.grid_encode_forward_counted(x, emb, offsets, enc, B, 3, 2, L, S, H, gridtype, align_corners, interp,
                                                          standalone=os.environ.get("FOC_GRID_PRECOUNT", "1") == "2") if training else None
        if ticket is None:
            _gridencoder.grid_encode_forward(x, emb, offsets, enc, B, 3, 2, L, S, H, None, gridtype, align_corners, interp)
        h = torch.empty(B, 16, device=x.device, dtype=torch.half)
        c = None
        if colour is not None:
            from ._lib import lib, ptr, stream_of, check
            cweights, ray_sh, T, c_layers, c_width, obj_feat = colour
            wc = _half_of(cweights)
            obj16 = obj_feat.detach().reshape(-1).half().contiguous() if obj_feat is not None else None
            c = torch.empty(B, c_width, device=x.device, dtype=torch.half)
            check(lib.foc_field_forward_train(ptr(enc), ptr(w), num_layers, ptr(ray_sh), int(T), ptr(wc), int(c_layers), 64, int(activation), B, ptr(h), ptr(c),
                                              int(c_width), ptr(obj16), stream_of(enc)), "field_forward_train")
        else:
            _ffmlp.ffmlp_forward_planar(enc, w, B, input_dim, 16, hidden_dim, num_layers, activation, output_activation, h)
        if training:
            ctx.save_for_backward(x, emb, w, offsets, enc)
            ctx.cfg = (enc_cfg, mlp_cfg, B, L)
            ctx.ticket = ticket
        if c is None:
            return h
        ctx.mark_non_differentiable(c)
        ctx.set_materialize_grads(False)          # c's "gradient" arrives as None instead of a zero-filled [B, c_width] tensor (6 us per step)
        return h, c

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_h, _grad_c=None):
        x, emb, w, offsets, enc = ctx.saved_tensors
        (S, H, gridtype, align_corners, interp), (input_dim, hidden_dim, num_layers, activation, output_activation), B, L = ctx.cfg
        if grad_h is None:                        # (set_materialize_grads(False): h took no part in the loss)
            return (None,) * 8
        grad_h = grad_h.contiguous().half()
        g_enc = torch.empty_like(enc)                           # [L,B,2]
        g_w = torch.empty_like(w)
        _ffmlp.ffmlp_backward_planar(grad_h, enc, w, B, input_dim, 16, hidden_dim, num_layers, activation, output_activation, True, g_enc, g_w)
        g_emb = torch.zeros_like(emb)
        _gridencoder.grid_encode_backward(g_enc, x, emb, offsets, g_emb, B, 3, 2, L, S, H, None, None, gridtype, align_corners, interp, grad_bl=False,
                                          precount=ctx.ticket)
        return None, g_emb, g_w, None, None, None, None, None


def infer_fusable(model):
    """Whole-field inference kernel (csrc/ffmlp.hip, k_nerf_infer): hash grid (D=3, C=2, 16 levels) -> 64-wide sigma net -> degree-4 SH
    + 15 geometry features (+ FOC's 16-wide encoded object feature, network_tcnn.py:611-640: 48-wide colour input) -> 64-wide colour net."""
    from .shencoder import SHEncoder
    enc, sn, cn = getattr(model, "encoder", None), getattr(model, "sigma_net", None), getattr(model, "color_net", None)
    obj = getattr(model, "uses_object_feature", False)
    return (field_fusable(enc, sn) and isinstance(cn, FFMLP) and isinstance(getattr(model, "encoder_dir", None), SHEncoder)
            and sn.input_dim == 32 and sn.hidden_dim == 64 and cn.hidden_dim == 64 and cn.input_dim == (48 if obj else 32) and cn.padded_output_dim == 16
            and getattr(model, "geo_feat_dim", 0) == 15 and (sn.num_layers, cn.num_layers) in ((2, 2), (2, 3), (3, 3))
            and (not obj or (getattr(model, "yolo_encoding_dim", 0) == 16 and sn.activation == 0))
            and sn.activation == cn.activation and os.environ.get("FOC_FUSED_INFER", "1") != "0")


@torch.no_grad()
def field_infer(model, xn, dirs, dir_div=1, dir_block=0, obj_feat=None):
    """xn [M,3] fp32 in [0,1] (already normalised), dirs [M / dir_div, 3] -> sigma [M] fp32, rgb [M,3] fp32 (no autograd).
    dir_block = 64: the rows stand in the block-interleaved order of `fixedstep.fixed_sample(..., ray_block=64)`.
    obj_feat [16]: the encoded object feature of an object-conditioned network (required iff `model.uses_object_feature`)."""
    from ._lib import lib, ptr, stream_of, check
    enc, sn, cn = model.encoder, model.sigma_net, model.color_net
    xn = xn.contiguous().float()
    dirs = dirs.contiguous().float()
    M = xn.shape[0]
    L = enc.offsets.shape[0] - 1
    emb, ws, wc = _half_of(enc.embeddings), _half_of(sn.weights), _half_of(cn.weights)
    planes = torch.empty(L, M, 2, device=xn.device, dtype=torch.half)
    _gridencoder.grid_encode_forward(xn, emb, enc.offsets, planes, M, 3, 2, L, float(np.log2(enc.per_level_scale)), enc.base_resolution, None,
                                     enc.gridtype_id, enc.align_corners, enc.interp_id)
    sigma = torch.empty(M, dtype=torch.float32, device=xn.device)
    rgb = torch.empty(M, 3, dtype=torch.float32, device=xn.device)
    obj16 = None
    if getattr(model, "uses_object_feature", False):
        if obj_feat is None:
            raise RuntimeError("field_infer: an object-conditioned network needs its encoded object feature")
        obj16 = obj_feat.detach().reshape(-1).half().contiguous()
        assert obj16.numel() == 16
    check(lib.foc_nerf_field_inference(ptr(planes), 1, ptr(dirs), int(dir_div), int(dir_block), dirs.shape[0], ptr(ws), sn.num_layers, ptr(wc), cn.num_layers, 64, sn.activation, M,
                                       ptr(sigma), ptr(rgb), ptr(obj16), stream_of(xn)), "nerf_field_inference")
    return sigma, rgb


def hashgrid_mlp(encoder, mlp, x, bound=1):
    """x [...,3] in [-bound, bound] -> [..., 16] half: mlp.forward_padded(encoder(x, bound))."""
    prefix = list(x.shape[:-1])
    xn = ((x + bound) / (2 * bound)).view(-1, 3)
    enc_cfg = (float(np.log2(encoder.per_level_scale)), encoder.base_resolution, encoder.gridtype_id, encoder.align_corners, encoder.interp_id)
    mlp_cfg = (mlp.input_dim, mlp.hidden_dim, mlp.num_layers, mlp.activation, mlp.output_activation)
    h = _hashgrid_mlp.apply(xn, encoder.embeddings, mlp.weights, encoder.offsets, enc_cfg, mlp_cfg, mlp.training and torch.is_grad_enabled())
    return h.view(prefix + [16])
