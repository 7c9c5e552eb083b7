"""Multiresolution hash-grid encoder — public surface of the reference's gridencoder/grid.py
(`grid_encode`, `GridEncoder`, :24-184), backed by libfocnerf_hip.so.

The kernels can write the encoding directly as [B, L*C] and read the incoming gradient in
that layout, so the two permute copies of the reference wrapper (grid.py:57, :75) are gone;
`FOCNERF_GRID_LBC=1` selects the reference's [L,B,C] kernel layout + permutes instead
(same values, used by the parity tests to cover both entry points).
"""
import os

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from .backend import _gridencoder as _backend

_gridtype_to_id = {'hash': 0, 'tiled': 1}
_interp_to_id = {'linear': 0, 'smoothstep': 1}


def _use_lbc():
    return os.environ.get("FOCNERF_GRID_LBC", "0") == "1"


class _grid_encode(Function):
    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0,
                align_corners=False, interpolation=0):
        # inputs [B,D] float in [0,1]; embeddings [rows,C]; offsets int32 [L+1]  ->  [B, L*C]
        inputs = inputs.contiguous()
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        C = embeddings.shape[1]
        S = np.log2(per_level_scale)      # float64 -> float32 at the ABI, as in the reference (grid.py:38)
        H = base_resolution

        # half-precision table under autocast when C is even (grid.py:41-44)
        if torch.is_autocast_enabled() and C % 2 == 0:
            embeddings = embeddings.to(torch.half)
        embeddings = embeddings.contiguous()

        dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=embeddings.dtype) if calc_grad_inputs else None
        lbc = _use_lbc()
        if lbc:
            outputs = torch.empty(L, B, C, device=inputs.device, dtype=embeddings.dtype)
            _backend.grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interpolation)
            outputs = outputs.permute(1, 0, 2).reshape(B, L * C)
        else:
            outputs = torch.empty(B, L * C, device=inputs.device, dtype=embeddings.dtype)
            unit = C * embeddings.element_size()
            if unit in (4, 8) and os.environ.get("FOCNERF_GRID_POINT_MAJOR", "0") != "1":
                # level-major kernel (one level's table in L2 at a time) + one transpose kernel: 0.45 vs 0.74 ms per 2 M incoherent points
                planes = torch.empty(L, B, C, device=inputs.device, dtype=embeddings.dtype)
                _backend.grid_encode_forward(inputs, embeddings, offsets, planes, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interpolation)
                _backend.planes_to_rows(planes, outputs, B, L, unit)
            else:
                _backend.grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interpolation,
                                             out_bl=True)

        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = [B, D, C, L, S, H, gridtype, interpolation]
        ctx.align_corners = align_corners
        ctx.lbc = lbc
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H, gridtype, interpolation = ctx.dims
        align_corners = ctx.align_corners

        grad = grad.to(embeddings.dtype)
        lbc_bwd = ctx.lbc or os.environ.get("FOCNERF_GRID_BWD_LBC", "0") == "1"
        if lbc_bwd:
            grad = grad.view(B, L, C).permute(1, 0, 2).contiguous()     # grid.py:75
        else:
            grad = grad.contiguous()
        grad_embeddings = torch.zeros_like(embeddings)
        grad_inputs = torch.zeros_like(inputs, dtype=embeddings.dtype) if dy_dx is not None else None
        _backend.grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype,
                                      align_corners, interpolation, grad_bl=not lbc_bwd)
        if dy_dx is not None:
            grad_inputs = grad_inputs.to(inputs.dtype)
        return grad_inputs, grad_embeddings, None, None, None, None, None, None, None


grid_encode = _grid_encode.apply


def level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners=False):
    """Rows per level, rounded up to a multiple of 8 (reference grid.py:117-128)."""
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


class GridEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None, gridtype='hash', align_corners=False, interpolation='linear'):
        super().__init__()
        if desired_resolution is not None:   # overrides per_level_scale (grid.py:101-102)
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))

        self.input_dim = input_dim
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.per_level_scale = per_level_scale
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype = gridtype
        self.gridtype_id = _gridtype_to_id[gridtype]
        self.interpolation = interpolation
        self.interp_id = _interp_to_id[interpolation]
        self.align_corners = align_corners
        self.max_params = 2 ** log2_hashmap_size

        offsets = level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners)
        self.register_buffer('offsets', torch.from_numpy(offsets))
        self.n_params = int(offsets[-1]) * level_dim
        self.embeddings = nn.Parameter(torch.empty(int(offsets[-1]), level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        std = 1e-4
        self.embeddings.data.uniform_(-std, std)

    def __repr__(self):
        return (f"GridEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"resolution={self.base_resolution} -> {int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))} "
                f"per_level_scale={self.per_level_scale:.4f} params={tuple(self.embeddings.shape)} gridtype={self.gridtype} "
                f"align_corners={self.align_corners} interpolation={self.interpolation}")

    def forward(self, inputs, bound=1):
        # inputs [..., input_dim] in [-bound, bound] -> [..., num_levels * level_dim]
        inputs = (inputs + bound) / (2 * bound)
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        outputs = grid_encode(inputs, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution, inputs.requires_grad,
                              self.gridtype_id, self.align_corners, self.interp_id)
        return outputs.view(prefix_shape + [self.output_dim])

    @torch.autocast(device_type="cuda", enabled=False)   # always fp32 (grid.py:163-164)
    def grad_total_variation(self, weight=1e-7, inputs=None, bound=1, B=1000000):
        D = self.input_dim
        C = self.embeddings.shape[1]
        L = self.offsets.shape[0] - 1
        S = np.log2(self.per_level_scale)
        H = self.base_resolution
        if inputs is None:
            inputs = torch.rand(B, self.input_dim, device=self.embeddings.device)
        else:
            inputs = (inputs + bound) / (2 * bound)
            inputs = inputs.view(-1, self.input_dim)
            B = inputs.shape[0]
        if self.embeddings.grad is None:
            raise ValueError('grad is None, should be called after loss.backward() and before optimizer.step()!')
        _backend.grad_total_variation(inputs.contiguous(), self.embeddings, self.embeddings.grad, self.offsets, weight, B, D, C, L, S, H,
                                      self.gridtype_id, self.align_corners)
