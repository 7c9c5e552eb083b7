"""Per-sample network glue as fused ops (csrc/head.hip): what nerf/network_ff.py:51-75 does with torch
expressions between the sigma network, the SH-encoded view direction and the colour network.

    sigma, cin = sample_head(h, dirs)      # sigma = trunc_exp(h[:,0]); cin = [SH4(dirs) | h[:,1:16] | 0]
    rgb = rgb_head(c)                      # sigmoid(c[:, :3]) (values rounded to fp16 like the half sigmoid), fp32 tensor

Used by `NeRFNetwork.forward` when its shapes are the FOC defaults (degree-4 SH, geo_feat_dim 15, 16-wide padded MLP
outputs); the torch expressions remain the fallback for every other configuration and the parity baseline in
tests/test_gpu_network.py.
"""
import torch
from torch.autograd import Function

from ._lib import lib, ptr, stream_of, check


class _sample_head(Function):
    @staticmethod
    def forward(ctx, h, dirs, obj_feat=None):
        """obj_feat: None -> 32-wide colour input; [16] half (one encoded object feature for all samples) -> 48-wide."""
        h = h.contiguous()
        dirs = dirs.contiguous().float()
        assert h.is_cuda and h.dtype == torch.float16 and h.dim() == 2 and h.shape[1] == 16
        M = h.shape[0]
        assert dirs.shape == (M, 3)
        width = 32 if obj_feat is None else 48
        if obj_feat is not None:
            obj_feat = obj_feat.detach().reshape(-1).half().contiguous()
            assert obj_feat.numel() == 16
        sigma = torch.empty(M, dtype=torch.float32, device=h.device)
        cin = torch.empty(M, width, dtype=torch.float16, device=h.device)
        check(lib.foc_sample_head_forward(ptr(h), ptr(dirs), M, ptr(sigma), ptr(cin), ptr(obj_feat), width, stream_of(h)), "sample_head_forward")
        ctx.save_for_backward(h)
        ctx.width = width
        return sigma, cin

    @staticmethod
    def backward(ctx, g_sigma, g_cin):
        (h,) = ctx.saved_tensors
        g_sigma = g_sigma.contiguous().float() if g_sigma is not None else None
        g_cin = g_cin.contiguous().half() if g_cin is not None else None
        grad_h = torch.empty_like(h)
        check(lib.foc_sample_head_backward(ptr(h), ptr(g_sigma), ptr(g_cin), h.shape[0], ptr(grad_h), ctx.width, stream_of(h)), "sample_head_backward")
        g_obj = None
        if ctx.width == 48 and ctx.needs_input_grad[2] and g_cin is not None:
            g_obj = g_cin[:, 31:47].float().sum(0)                 # one feature vector feeds every sample
        return grad_h, None, g_obj


class _rgb_head(Function):
    @staticmethod
    def forward(ctx, c):
        c = c.contiguous()
        assert c.is_cuda and c.dtype == torch.float16 and c.dim() == 2 and c.shape[1] == 16
        rgb = torch.empty(c.shape[0], 3, dtype=torch.float32, device=c.device)
        check(lib.foc_rgb_head_forward(ptr(c), c.shape[0], ptr(rgb), stream_of(c)), "rgb_head_forward")
        ctx.save_for_backward(c)
        return rgb

    @staticmethod
    def backward(ctx, g_rgb):
        (c,) = ctx.saved_tensors
        g_rgb = g_rgb.contiguous().float()
        grad_c = torch.empty_like(c)
        check(lib.foc_rgb_head_backward(ptr(c), ptr(g_rgb), c.shape[0], ptr(grad_c), stream_of(c)), "rgb_head_backward")
        return grad_c


sample_head = _sample_head.apply
rgb_head = _rgb_head.apply
