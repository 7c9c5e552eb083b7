"""Frequency encoding [x, sin(2^f x), cos(2^f x)]_{f < degree} on the GPU (csrc/freqencoder.hip).

Drop-in for the reference's freqencoder/freq.py: `freq_encode(inputs, degree, output_dim)` and `FreqEncoder(input_dim, degree)` with the
same attributes; always fp32 (freq.py:17). The backward pass needs the encoding only (d sin = cos, d cos = -sin are already in it).
"""
import torch
import torch.nn as nn

from ._autograd import AmpOp, on_gpu, rows, unrows
from .backend import _freqencoder as _kernels


def encoded_width(input_dim, degree):
    return input_dim * (1 + 2 * degree)


class FrequencyEncoding(AmpOp):
    cast = torch.float32

    @staticmethod
    def run(ctx, points, degree, width):
        points = on_gpu(points).contiguous()
        n, dim = points.shape
        encoded = points.new_empty(n, width)
        _kernels.freq_encode_forward(points, n, dim, degree, width, encoded)
        ctx.save_for_backward(encoded)
        ctx.geometry = (n, dim, degree, width)
        return encoded

    @staticmethod
    def grad(ctx, upstream):
        (encoded,) = ctx.saved_tensors
        n, dim, degree, width = ctx.geometry
        d_points = encoded.new_zeros(n, dim)
        _kernels.freq_encode_backward(upstream.contiguous(), encoded, n, dim, degree, width, d_points)
        return d_points, None, None


freq_encode = FrequencyEncoding.apply


class FreqEncoder(nn.Module):
    """`output_dim = input_dim * (1 + 2 * degree)`; accepts any leading shape."""

    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim, self.degree = input_dim, degree
        self.output_dim = encoded_width(input_dim, degree)

    def extra_repr(self):
        return f"input_dim={self.input_dim}, degree={self.degree}, output_dim={self.output_dim}"

    def forward(self, inputs, **_unused):
        flat, lead = rows(inputs, self.input_dim)
        return unrows(freq_encode(flat, self.degree, self.output_dim), lead)
