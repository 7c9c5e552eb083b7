"""NeRFNetwork on GridEncoder + FFMLP — same topology and method signatures as the reference's
nerf/network_ff.py:10-148 (sigma-net 32->64->64->16 with trunc_exp on channel 0, colour-net
[SH16 | geo_feat 15 | pad 1] = 32->64->64->64->16 with sigmoid), built on this repo's ops.
The reference file cannot be constructed from its own tree (it imports the missing `encoding`
module); with focnerf_amd/dropin on PYTHONPATH it can.
"""
import os

import torch

from .activation import trunc_exp
from .encoding import get_encoder
from .ffmlp import FFMLP
from .renderer import NeRFRenderer


class NeRFNetwork(NeRFRenderer):
    def __init__(self, encoding="hashgrid", encoding_dir="sphere_harmonics", num_layers=2, hidden_dim=64, geo_feat_dim=15,
                 num_layers_color=3, hidden_dim_color=64, bound=1, **kwargs):
        super().__init__(bound, **kwargs)
        self.num_layers = num_layers
        self.hidden_dim = hidden_dim
        self.geo_feat_dim = geo_feat_dim
        self.encoder, self.in_dim = get_encoder(encoding, desired_resolution=2048 * bound)
        self.sigma_net = FFMLP(input_dim=self.in_dim, output_dim=1 + self.geo_feat_dim, hidden_dim=self.hidden_dim, num_layers=self.num_layers)

        self.num_layers_color = num_layers_color
        self.hidden_dim_color = hidden_dim_color
        self.encoder_dir, self.in_dim_color = get_encoder(encoding_dir)
        self.in_dim_color += self.geo_feat_dim + 1   # pad to 32 (network_ff.py:44)
        self.color_net = FFMLP(input_dim=self.in_dim_color, output_dim=3, hidden_dim=self.hidden_dim_color, num_layers=self.num_layers_color)

    def _fused_head_ok(self, x):
        """The fused glue of csrc/head.hip covers the FOC default shapes: degree-4 SH directions, 15 geometry features,
        both MLPs on FFMLP with 16-wide padded outputs, fp16 autocast, flat [M,3] CUDA inputs."""
        from .shencoder import SHEncoder
        return (x.is_cuda and x.dim() == 2 and torch.is_autocast_enabled() and self.geo_feat_dim == 15
                and isinstance(self.encoder_dir, SHEncoder) and getattr(self.encoder_dir, "degree", 0) == 4
                and isinstance(self.sigma_net, FFMLP) and isinstance(self.color_net, FFMLP) and self.in_dim_color == 32
                and os.environ.get("FOC_FUSED_HEAD", "1") != "0")

    def forward(self, x, d):
        if self._fused_head_ok(x):
            from .head import sample_head, rgb_head
            from .field import field_fusable, hashgrid_mlp, infer_fusable, field_infer
            if not torch.is_grad_enabled() and infer_fusable(self):
                # inference: encoder planes -> one kernel for both networks and the glue between them
                return field_infer(self, (x + self.bound) / (2 * self.bound), d)
            if field_fusable(self.encoder, self.sigma_net):
                h = hashgrid_mlp(self.encoder, self.sigma_net, x, self.bound)          # [M,16] half, encoding kept in [L,B,C]
            else:
                h = self.sigma_net.forward_padded(self.encoder(x, bound=self.bound))
            sigma, cin = sample_head(h, d)
            return sigma, rgb_head(self.color_net.forward_padded(cin))
        x = self.encoder(x, bound=self.bound)
        h = self.sigma_net(x)
        sigma = trunc_exp(h[..., 0])
        geo_feat = h[..., 1:]
        d = self.encoder_dir(d)
        p = torch.zeros_like(geo_feat[..., :1])
        h = torch.cat([d.to(geo_feat.dtype), geo_feat, p], dim=-1)
        h = self.color_net(h)
        rgb = torch.sigmoid(h)
        return sigma, rgb

    def density(self, x):
        from .field import field_fusable, hashgrid_mlp
        if x.is_cuda and x.dim() == 2 and torch.is_autocast_enabled() and field_fusable(self.encoder, self.sigma_net):
            h = hashgrid_mlp(self.encoder, self.sigma_net, x, self.bound)       # same values; encoding kept in [L,B,C] planes
        else:
            h = self.sigma_net(self.encoder(x, bound=self.bound))
        sigma = trunc_exp(h[..., 0])
        geo_feat = h[..., 1:]
        return {'sigma': sigma, 'geo_feat': geo_feat}

    def color(self, x, d, mask=None, geo_feat=None, **kwargs):
        if mask is not None:
            rgbs = torch.zeros(mask.shape[0], 3, dtype=x.dtype, device=x.device)
            if not mask.any():
                return rgbs
            x = x[mask]
            d = d[mask]
            geo_feat = geo_feat[mask]
        d = self.encoder_dir(d)
        p = torch.zeros_like(geo_feat[..., :1])
        h = torch.cat([d.to(geo_feat.dtype), geo_feat, p], dim=-1)
        h = self.color_net(h)
        h = torch.sigmoid(h)
        if mask is not None:
            rgbs[mask] = h.to(rgbs.dtype)
        else:
            rgbs = h
        return rgbs

    def run(self, rays_o, rays_d, yolo_details=None, fused=False, **kwargs):
        """`fused=True` routes the fixed-step path through csrc/fixedstep.hip (same image, depth and gradients as the
        torch code of NeRFRenderer.run, which stays the default)."""
        if fused and kwargs.get("upsample_steps", 0) == 0 and self.bg_radius <= 0:
            from .fixedstep import render_fixed_steps
            kwargs.pop("upsample_steps", None)
            return render_fixed_steps(self, rays_o, rays_d, yolo_details=yolo_details, **kwargs)
        return super().run(rays_o, rays_d, yolo_details, **kwargs)

    def get_params(self, lr):
        return [
            {'params': self.encoder.parameters(), 'lr': lr},
            {'params': self.sigma_net.parameters(), 'lr': lr},
            {'params': self.encoder_dir.parameters(), 'lr': lr},
            {'params': self.color_net.parameters(), 'lr': lr},
        ]
