"""Fully fused fp16 MLP on the matrix cores (csrc/ffmlp.hip).

Drop-in for the reference's ffmlp/ffmlp.py: `ffmlp_forward(...)` with its positional arguments and `FFMLP(input_dim, output_dim,
hidden_dim, num_layers, activation)` with its attributes, its single `weights` parameter in the reference's blob layout
([hidden x input] | (num_layers - 1) x [hidden x hidden] | [padded_output x hidden], every matrix row-major with the output neuron
as the row, ffmlp.py:120-122) and its seed-42 U(+-sqrt(3 / hidden)) initialisation (ffmlp.py:141-144).

Differences that a caller cannot observe in the results:
  * no batch padding: the reference copies the input into a zero-padded batch of k * 128 rows (ffmlp.py:157-159); the kernels handle a
    ragged last tile, so that copy is not made;
  * training keeps no activations for the shapes the single-pass backward kernel serves (`single_pass_backward`): it re-evaluates them on
    chip from the inputs (same MFMA sequence, same bits). FOC_MLP_RECOMPUTE=0 stores them as the reference does (ffmlp.py:31);
  * accumulation is fp32 on the matrix cores (the reference accumulates in fp16).
The reference module's stray `from turtle import ...` (ffmlp.py:2) has no counterpart.
"""
import math
import os

import torch
import torch.nn as nn

from ._autograd import AmpOp
from .backend import _ffmlp as _kernels

ACTIVATIONS = {'relu': 0, 'exponential': 1, 'sine': 2, 'sigmoid': 3, 'squareplus': 4, 'softplus': 5}
NO_ACTIVATION = 6
SUPPORTED_HIDDEN = (16, 32, 64, 128, 256)       # what the reference accepts (ffmlp.py:112; 256 runs layer by layer, csrc/ffmlp_wide.hip)


def convert_activation(name):
    return ACTIVATIONS.get(name, NO_ACTIVATION)


def single_pass_backward(input_dim, hidden_dim, num_layers, activation=0):
    """Shapes served by k_mlp_bwd_fused (activation gradients, weight gradients and input gradients in one pass): ReLU / no activation —
    every NeRF network; the reference's other hidden activations (exponential, sine, sigmoid, squareplus, softplus, utils.h:424-589) take
    its own data flow, stored activations and a [layers, B, hidden] gradient buffer."""
    return (activation in (ACTIVATIONS['relu'], NO_ACTIVATION) and hidden_dim <= 64 and input_dim <= 64 and 2 <= num_layers <= 4
            and _fused_backward_switch())


def _fused_backward_switch():
    from ._lib import get_option
    return get_option("FOC_MLP_BWD_FUSED") != 0


_fused_backward_ok = single_pass_backward       # name used by focnerf_amd.field


def _keeps_activations(input_dim, hidden_dim, num_layers, activation=0):
    return not (single_pass_backward(input_dim, hidden_dim, num_layers, activation) and os.environ.get("FOC_MLP_RECOMPUTE", "1") != "0")


class FusedMLP(AmpOp):
    """rows [B, input_dim] half, weight blob half -> [B, output_dim (padded to 16)] half."""
    cast = torch.half

    @staticmethod
    def run(ctx, x, blob, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, inference, want_dx):
        x, blob = x.contiguous(), blob.contiguous()
        n = x.shape[0]
        net = (input_dim, output_dim, hidden_dim, num_layers, activation, output_activation)
        y = x.new_empty(n, output_dim)
        if inference:
            # activations stay in registers: no [B, hidden] scratch (the reference allocates one, ffmlp.py:41) — except at hidden 256, where one
            # matrix fills the LDS and the layers run one launch each, in place in that buffer
            _kernels.ffmlp_inference(x, blob, n, *net, x.new_empty(n, hidden_dim) if hidden_dim > 128 else None, y)
            return y
        kept = x.new_empty(num_layers, n, hidden_dim) if _keeps_activations(input_dim, hidden_dim, num_layers, activation) else None
        _kernels.ffmlp_forward(x, blob, n, *net, kept, y)
        ctx.save_for_backward(x, blob, kept)
        ctx.net, ctx.want_dx = net, want_dx
        return y

    @staticmethod
    def grad(ctx, dy):
        x, blob, kept = ctx.saved_tensors
        input_dim, _, hidden_dim, num_layers = ctx.net[:4]
        n = dy.shape[0]
        dx = torch.empty_like(x) if ctx.want_dx else x.new_zeros(1)
        d_blob = torch.empty_like(blob)
        # the [num_layers, B, hidden] gradient buffer of ffmlp.py:73 exists only for the two-kernel form (hidden 128, > 4 layers, wide inputs)
        scratch = None if single_pass_backward(input_dim, hidden_dim, num_layers, ctx.net[4]) else x.new_empty(num_layers, n, hidden_dim)
        _kernels.ffmlp_backward(dy.contiguous(), x, blob, kept, n, *ctx.net, ctx.want_dx, scratch, dx, d_blob)
        return (dx if ctx.want_dx else None, d_blob) + (None,) * 8


def ffmlp_forward(inputs, weights, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, inference=False,
                  calc_grad_inputs=False):
    """Positional signature of the reference's `ffmlp_forward = _ffmlp_forward.apply` (ffmlp.py:25-26, :97)."""
    return FusedMLP.apply(inputs, weights, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, inference,
                          calc_grad_inputs)


class FFMLP(nn.Module):
    def __init__(self, input_dim, output_dim, hidden_dim, num_layers, activation='relu'):
        super().__init__()
        if hidden_dim not in SUPPORTED_HIDDEN:
            raise AssertionError(f"FFMLP only support hidden_dim in [16, 32, 64, 128, 256], but got {hidden_dim}")
        if input_dim <= 0 or input_dim % 16:
            raise AssertionError(f"FFMLP input_dim should be 16 * m (m  > 0), but got {input_dim}")
        if output_dim > 16:
            raise AssertionError(f"FFMLP current only supports output dim <= 16, but got {output_dim}")
        if num_layers < 2:
            raise AssertionError(f"FFMLP num_layers should be larger than 2 (3 matmuls), but got {num_layers}")
        self.input_dim, self.output_dim, self.hidden_dim, self.num_layers = input_dim, output_dim, hidden_dim, num_layers
        self.activation, self.output_activation = convert_activation(activation), convert_activation('none')
        self.tensorcore_width = 16
        self.padded_output_dim = -(-output_dim // 16) * 16
        self.num_parameters = hidden_dim * (input_dim + hidden_dim * (num_layers - 1) + self.padded_output_dim)
        self.weights = nn.Parameter(torch.zeros(self.num_parameters))
        self.reset_parameters()
        _kernels.allocate_splitk(num_layers + 1)              # kept for the call sequence; weight gradients need no side streams here

    def cleanup(self):
        _kernels.free_splitk()

    def extra_repr(self):
        return (f"{self.input_dim} -> " + " -> ".join([str(self.hidden_dim)] * self.num_layers) + f" -> {self.output_dim}, "
                f"activation={self.activation}, parameters={self.num_parameters}")

    def reset_parameters(self):
        torch.manual_seed(42)
        bound = math.sqrt(3 / self.hidden_dim)
        self.weights.data.uniform_(-bound, bound)

    def forward_padded(self, inputs):
        """The kernel's full [B, 16] result (columns >= output_dim belong to the padding neurons), without the slice copy: the fused
        render path's composite kernel reads the 16-wide rows as they are."""
        return ffmlp_forward(inputs, self.weights, self.input_dim, self.padded_output_dim, self.hidden_dim, self.num_layers, self.activation,
                             self.output_activation, not self.training, inputs.requires_grad)

    def forward(self, inputs):
        y = self.forward_padded(inputs)
        return y if self.padded_output_dim == self.output_dim else y[:, :self.output_dim]
