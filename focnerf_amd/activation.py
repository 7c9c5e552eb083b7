"""`trunc_exp`: the density activation of the NeRF networks.

Behaviour of the reference's activation.py:5-17: exp evaluated in fp32 whatever the autocast dtype; the derivative is taken at the
argument clamped to [-15, 15], so that one overflowing logit cannot turn the whole gradient into inf.
(The fused render paths evaluate the same two expressions inside their kernels: csrc/fixedstep.hip, csrc/head.hip.)
"""
import torch

from ._autograd import AmpOp

_GRAD_CLAMP = 15.0


class TruncatedExp(AmpOp):
    cast = torch.float32

    @staticmethod
    def run(ctx, logits):
        ctx.save_for_backward(logits)
        return logits.exp()

    @staticmethod
    def grad(ctx, upstream):
        (logits,) = ctx.saved_tensors
        return upstream * logits.clamp(min=-_GRAD_CLAMP, max=_GRAD_CLAMP).exp()


trunc_exp = TruncatedExp.apply
