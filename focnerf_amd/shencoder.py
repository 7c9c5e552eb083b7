"""Degree-4 real spherical-harmonics direction encoder (16 outputs).

The reference imports `encoding.get_encoder('sphere_harmonics')` (nerf/network_ff.py:5,43) but
ships neither encoding.py nor shencoder/ (SURVEY.md H2). This is the published degree-4 basis
used by torch-ngp / instant-ngp, written with torch ops — PARITY UNPINNED (nothing in the
reference tree to check it against; orthonormality is tested instead).
"""
import torch
import torch.nn as nn


def sh_encode_deg4(d):
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    xy, xz, yz = x * y, x * z, y * z
    x2, y2, z2 = x * x, y * y, z * z
    out = torch.stack([
        torch.full_like(x, 0.28209479177387814),
        -0.48860251190291987 * y,
        0.48860251190291987 * z,
        -0.48860251190291987 * x,
        1.0925484305920792 * xy,
        -1.0925484305920792 * yz,
        0.94617469575755997 * z2 - 0.31539156525251999,
        -1.0925484305920792 * xz,
        0.54627421529603959 * x2 - 0.54627421529603959 * y2,
        0.59004358992664352 * y * (-3.0 * x2 + y2),
        2.8906114426405538 * xy * z,
        0.45704579946446572 * y * (1.0 - 5.0 * z2),
        0.3731763325901154 * z * (5.0 * z2 - 3.0),
        0.45704579946446572 * x * (1.0 - 5.0 * z2),
        1.4453057213202769 * z * (x2 - y2),
        0.59004358992664352 * x * (-x2 + 3.0 * y2),
    ], dim=-1)
    return out


class SHEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        assert input_dim == 3 and degree == 4, "only the degree-4 basis used by the NeRF networks is provided"
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = degree ** 2

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"

    def forward(self, inputs, **kwargs):
        inputs = inputs.float()
        if inputs.is_cuda and not (inputs.requires_grad and torch.is_grad_enabled()):
            # one kernel (csrc/head.hip, same expressions) instead of 17; directions carry no gradient in the NeRF networks
            from ._lib import lib, ptr, stream_of, check
            flat = inputs.contiguous().view(-1, 3)
            out = torch.empty(flat.shape[0], 16, dtype=torch.float32, device=inputs.device)
            check(lib.foc_sh_encode(ptr(flat), flat.shape[0], ptr(out), stream_of(flat)), "sh_encode")
            return out.view(*inputs.shape[:-1], 16)
        return sh_encode_deg4(inputs)
