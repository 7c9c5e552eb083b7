"""Multiresolution hash-grid encoder on the GPU (csrc/gridencoder.hip).

Drop-in for the reference's gridencoder/grid.py: `grid_encode(...)` with its positional arguments and `GridEncoder` with its constructor
arguments, attributes, parameter / buffer names (`embeddings`, `offsets`) and `grad_total_variation`.

How a call is served (the reference always runs its [L,B,C] kernel and permutes, grid.py:47,57,75):
  * forward  — level-major kernel into [L,B,C] planes + one transpose kernel to the [B, L*C] rows the caller gets; the point-major
               kernel that writes rows directly sits behind FOCNERF_GRID_POINT_MAJOR=1, the reference's layout + torch permute behind
               FOCNERF_GRID_LBC=1 (the parity tests run all three);
  * backward — D = 3, C = 2 tables: partition + on-chip accumulation (no scattered atomics), reading the incoming gradient as
               [B, L*C] rows; other shapes: the atomic kernel (FOCNERF_GRID_ATOMIC=1 forces it).
"""
import os
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn as nn

from ._autograd import AmpOp, rows, unrows
from .backend import _gridencoder as _kernels

GRID_TYPES = {"hash": 0, "tiled": 1}
INTERPOLATIONS = {"linear": 0, "smoothstep": 1}


def _flag(name):
    return os.environ.get(name, "0") == "1"


@dataclass(frozen=True)
class GridSpec:
    """Everything the kernels need besides the tensors. `log2_scale` is what the reference passes as S (grid.py:38)."""
    log2_scale: float
    base_resolution: int
    gridtype: int = 0
    align_corners: bool = False
    interpolation: int = 0

    def tail(self):
        return self.gridtype, self.align_corners, self.interpolation


class HashGridEncode(AmpOp):
    """points [B,D] in [0,1], table [rows,C], offsets int32 [L+1] -> [B, L*C]. Under autocast the table is read as fp16 when C is even
    (grid.py:41-44) and the result has the table's dtype."""

    @staticmethod
    def run(ctx, points, table, offsets, spec, want_dx, grad_mode=True):
        points = points.contiguous()
        n, dim = points.shape
        levels, chans = offsets.numel() - 1, table.shape[1]
        if torch.is_autocast_enabled() and chans % 2 == 0:
            table = table.to(torch.half)
        table = table.contiguous()
        like = dict(device=points.device, dtype=table.dtype)
        dy_dx = torch.empty(n, levels * dim * chans, **like) if want_dx else None
        shape = (n, dim, chans, levels, spec.log2_scale, spec.base_resolution)

        reference_layout = _flag("FOCNERF_GRID_LBC")
        unit = chans * table.element_size()
        ticket = None
        if reference_layout:
            planes = torch.empty(levels, n, chans, **like)
            _kernels.grid_encode_forward(points, table, offsets, planes, *shape, dy_dx, *spec.tail())
            encoded = planes.permute(1, 0, 2).reshape(n, levels * chans)
        elif unit in (4, 8) and not _flag("FOCNERF_GRID_POINT_MAJOR"):
            planes = torch.empty(levels, n, chans, **like)
            encoded = torch.empty(n, levels * chans, **like)
            # a forward whose table will receive a gradient lets the backward's count pass ride in its launch (the kernels' own callers do
            # the same, focnerf_amd/field.py): the backward then starts at its scatter — 0.10 ms of a 2 M-point call. `needs_input_grad`
            # alone stays True under torch.no_grad() for an nn.Parameter table: an evaluation call must neither run the count pass, nor
            # create the backward's scratch, nor take a precount ticket away from a pending training forward. `grad_mode` is the caller's
            # torch.is_grad_enabled() (inside a Function's forward it always reads False)
            if dy_dx is None and dim == 3 and chans == 2 and ctx.needs_input_grad[1] and grad_mode and n:
                ticket = _kernels.grid_encode_forward_counted(points, table, offsets, planes, *shape, *spec.tail())
            if ticket is None:
                _kernels.grid_encode_forward(points, table, offsets, planes, *shape, dy_dx, *spec.tail())
            _kernels.planes_to_rows(planes, encoded, n, levels, unit)
        else:
            encoded = torch.empty(n, levels * chans, **like)
            _kernels.grid_encode_forward(points, table, offsets, encoded, *shape, dy_dx, *spec.tail(), out_bl=True)

        ctx.save_for_backward(points, table, offsets, dy_dx)
        ctx.call = (shape, spec, reference_layout)
        ctx.ticket = ticket
        return encoded

    @staticmethod
    def grad(ctx, upstream):
        points, table, offsets, dy_dx = ctx.saved_tensors
        shape, spec, reference_layout = ctx.call
        n, dim, chans, levels = shape[:4]
        upstream = upstream.to(table.dtype)
        planes_in = reference_layout or _flag("FOCNERF_GRID_BWD_LBC")
        unit = chans * table.element_size()
        if planes_in:
            upstream = upstream.view(n, levels, chans).permute(1, 0, 2).contiguous()
        elif unit in (4, 8) and dim == 3 and chans == 2 and n:
            # the binned backward reads planes faster than rows: one transpose kernel, then the [L,B,C] form (FOCNERF_GRID_BWD_ROWS=1: rows as they are)
            if not _flag("FOCNERF_GRID_BWD_ROWS"):
                planes = torch.empty(levels, n, chans, device=upstream.device, dtype=table.dtype)
                _kernels.rows_to_planes(upstream.contiguous(), planes, n, levels, unit)
                upstream, planes_in = planes, True
            else:
                upstream = upstream.contiguous()
        else:
            upstream = upstream.contiguous()
        d_table = torch.zeros_like(table)
        d_points = torch.zeros_like(points, dtype=table.dtype) if dy_dx is not None else None
        _kernels.grid_encode_backward(upstream, points, table, offsets, d_table, *shape, dy_dx, d_points, *spec.tail(), grad_bl=not planes_in,
                                      precount=ctx.ticket)
        if d_points is not None:
            d_points = d_points.to(points.dtype)
        return d_points, d_table, None, None, None, None


def grid_encode(inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0, align_corners=False,
                interpolation=0):
    """Positional signature of the reference's `grid_encode = _grid_encode.apply` (grid.py:27-28, :92)."""
    spec = GridSpec(float(np.log2(per_level_scale)), int(base_resolution), int(gridtype), bool(align_corners), int(interpolation))
    return HashGridEncode.apply(inputs, embeddings, offsets, spec, bool(calc_grad_inputs), torch.is_grad_enabled())


def level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners=False):
    """First row of every level (+ the total): a level holds min(2^log2_hashmap_size, cells^input_dim) rows, padded to a multiple
    of 8, with cells = ceil(base * scale^level) (+1 unless align_corners) — the table layout of grid.py:117-128."""
    cap = 2 ** log2_hashmap_size
    sizes = []
    for level in range(num_levels):
        cells = int(np.ceil(base_resolution * per_level_scale ** level)) + (0 if align_corners else 1)
        sizes.append(-(-min(cap, cells ** input_dim) // 8) * 8)
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)


class GridEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None, gridtype='hash', align_corners=False, interpolation='linear'):
        super().__init__()
        if desired_resolution is not None:
            # geometric growth from base_resolution to desired_resolution over the levels (takes precedence, grid.py:101-102)
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim, self.num_levels, self.level_dim = input_dim, num_levels, level_dim
        self.per_level_scale, self.base_resolution = per_level_scale, base_resolution
        self.log2_hashmap_size, self.max_params = log2_hashmap_size, 2 ** log2_hashmap_size
        self.output_dim = num_levels * level_dim
        self.gridtype, self.gridtype_id = gridtype, GRID_TYPES[gridtype]
        self.interpolation, self.interp_id = interpolation, INTERPOLATIONS[interpolation]
        self.align_corners = align_corners

        offsets = level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners)
        total_rows = int(offsets[-1])
        self.register_buffer('offsets', torch.from_numpy(offsets))
        self.n_params = total_rows * level_dim
        self.embeddings = nn.Parameter(torch.empty(total_rows, level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        self.embeddings.data.uniform_(-1e-4, 1e-4)            # grid.py:139-141

    def extra_repr(self):
        finest = int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))
        return (f"input_dim={self.input_dim}, levels={self.num_levels} x {self.level_dim}, resolution {self.base_resolution}..{finest} "
                f"(x{self.per_level_scale:.4f}), table={tuple(self.embeddings.shape)}, {self.gridtype}/{self.interpolation}, "
                f"align_corners={self.align_corners}")

    def _unit_cube(self, x, bound):
        return (x + bound) / (2 * bound)

    def forward(self, inputs, bound=1):
        """inputs [..., input_dim] in [-bound, bound] -> [..., num_levels * level_dim]."""
        flat, lead = rows(self._unit_cube(inputs, bound), self.input_dim)
        encoded = grid_encode(flat, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution, flat.requires_grad,
                              self.gridtype_id, self.align_corners, self.interp_id)
        return unrows(encoded, lead)

    @torch.autocast(device_type="cuda", enabled=False)
    def grad_total_variation(self, weight=1e-7, inputs=None, bound=1, B=1000000):
        """Adds the total-variation gradient at `inputs` (default: B uniform samples) to `embeddings.grad`, in fp32 (grid.py:163-184):
        call it between loss.backward() and optimizer.step()."""
        if self.embeddings.grad is None:
            raise ValueError('grad is None, should be called after loss.backward() and before optimizer.step()!')
        if inputs is None:
            at = torch.rand(B, self.input_dim, device=self.embeddings.device)
        else:
            at, _ = rows(self._unit_cube(inputs, bound), self.input_dim)
        _kernels.grad_total_variation(at.contiguous(), self.embeddings, self.embeddings.grad, self.offsets, weight, at.shape[0], self.input_dim,
                                      self.embeddings.shape[1], self.offsets.numel() - 1, np.log2(self.per_level_scale), self.base_resolution,
                                      self.gridtype_id, self.align_corners)
