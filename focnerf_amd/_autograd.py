"""Plumbing shared by the operator modules (activation, freqencoder, gridencoder, ffmlp, raymarching).

`AmpOp` is a torch.autograd.Function whose subclasses state their autocast policy as data and write `run` / `grad`:

    class Op(AmpOp):
        cast = torch.float32            # inputs are cast to this dtype under autocast; KEEP = left alone, autocast state recorded

        @staticmethod
        def run(ctx, ...): ...
        @staticmethod
        def grad(ctx, *upstream): ...

The reference decorates every op by hand with torch.cuda.amp.custom_fwd / custom_bwd (e.g. raymarching.py:241, grid.py:26,
ffmlp.py:23, freq.py:17); the policy per op is kept, the mechanism lives here once.
"""
import torch
from torch.amp import custom_bwd, custom_fwd
from torch.autograd import Function

KEEP = object()


class AmpOp(Function):
    cast = KEEP

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        run, grad = cls.__dict__.get("run"), cls.__dict__.get("grad")
        if run is None:
            return
        fwd = custom_fwd(device_type="cuda") if cls.cast is KEEP else custom_fwd(device_type="cuda", cast_inputs=cls.cast)
        cls.forward = staticmethod(fwd(run.__func__))
        if grad is not None:
            cls.backward = staticmethod(custom_bwd(device_type="cuda")(grad.__func__))


def rows(x, width):
    """[..., width] -> ([n, width], leading shape): the ops work on flat row lists, the modules accept any leading shape."""
    lead = tuple(x.shape[:-1])
    return x.reshape(-1, width), lead


def unrows(y, lead):
    return y.reshape(*lead, y.shape[-1])


def on_gpu(*tensors):
    """Host tensors are moved to the current device (the reference wrappers do the same: `if not x.is_cuda: x = x.cuda()`)."""
    moved = tuple(t if (t is None or t.is_cuda) else t.cuda() for t in tensors)
    return moved[0] if len(moved) == 1 else moved
