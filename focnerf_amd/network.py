"""The NeRF field FOC trains per object: hash-grid encoder -> density MLP -> (direction encoding, geometry features) -> colour MLP.

Same constructor arguments, sub-module names (`encoder`, `sigma_net`, `encoder_dir`, `color_net` — hence the same state_dict keys),
methods (`forward`, `density`, `color`, `get_params`) and numerics as the reference's nerf/network_ff.py:10-148:
    h     = sigma_net(encoder(x))                     32 -> 64 -> 64 -> 16
    sigma = trunc_exp(h[:, 0]);  geo = h[:, 1:16]
    rgb   = sigmoid(color_net([SH16(d) | geo | 0]))    32 -> 64 -> 64 -> 64 -> 3 (padded to 16)
The reference file itself cannot be constructed from its own tree (it imports an `encoding` module the tree lacks); with
focnerf_amd/dropin on PYTHONPATH it can (tests/test_dropin.py).

Under fp16 autocast on flat [M,3] GPU inputs the glue between the kernels (trunc_exp, SH, concatenation + padding, sigmoid) runs as
fused kernels too (csrc/head.hip; FOC_FUSED_HEAD=0 keeps the torch expressions), and without autograd the whole field is one kernel
after the encoder (csrc/ffmlp.hip, k_nerf_infer).
"""
import os

import torch

from .activation import trunc_exp
from .encoding import get_encoder
from .ffmlp import FFMLP
from .renderer import NeRFRenderer


class NeRFNetwork(NeRFRenderer):
    def __init__(self, encoding="hashgrid", encoding_dir="sphere_harmonics", num_layers=2, hidden_dim=64, geo_feat_dim=15, num_layers_color=3,
                 hidden_dim_color=64, bound=1, **kwargs):
        super().__init__(bound, **kwargs)
        self.num_layers, self.hidden_dim, self.geo_feat_dim = num_layers, hidden_dim, geo_feat_dim
        self.num_layers_color, self.hidden_dim_color = num_layers_color, hidden_dim_color
        # density branch: finest level of the grid resolves 1/2048 of a unit box whatever the bound
        self.encoder, self.in_dim = get_encoder(encoding, desired_resolution=2048 * bound)
        self.sigma_net = FFMLP(self.in_dim, 1 + geo_feat_dim, hidden_dim, num_layers)
        # colour branch: direction encoding + geometry features, one zero column to reach a multiple of 16 (network_ff.py:44)
        self.encoder_dir, dir_width = get_encoder(encoding_dir)
        self.in_dim_color = dir_width + geo_feat_dim + 1
        self.color_net = FFMLP(self.in_dim_color, 3, hidden_dim_color, num_layers_color)

    # ---- torch expressions (any dtype / device / shape)
    def _colour_input(self, d, geo_feat):
        return torch.cat([self.encoder_dir(d).to(geo_feat.dtype), geo_feat, torch.zeros_like(geo_feat[..., :1])], dim=-1)

    def _shade(self, d, geo_feat):
        return torch.sigmoid(self.color_net(self._colour_input(d, geo_feat)))

    # ---- fused kernels
    def _fused_head_ok(self, x):
        """csrc/head.hip serves FOC's shapes: degree-4 SH, 15 geometry features, both networks FFMLPs with 16-wide padded outputs, fp16
        autocast, flat [M,3] GPU inputs."""
        from .shencoder import SHEncoder
        return (x.is_cuda and x.dim() == 2 and torch.is_autocast_enabled() and self.geo_feat_dim == 15 and self.in_dim_color == 32
                and isinstance(self.encoder_dir, SHEncoder) and getattr(self.encoder_dir, "degree", 0) == 4
                and isinstance(self.sigma_net, FFMLP) and isinstance(self.color_net, FFMLP) and os.environ.get("FOC_FUSED_HEAD", "1") != "0")

    def _geometry_rows(self, x):
        """[M,3] -> [M,16] half: the density network's padded output, the encoding kept in the encoder's [L,B,C] planes when possible."""
        from .field import field_fusable, hashgrid_mlp
        if field_fusable(self.encoder, self.sigma_net):
            return hashgrid_mlp(self.encoder, self.sigma_net, x, self.bound)
        return self.sigma_net.forward_padded(self.encoder(x, bound=self.bound))

    def forward(self, x, d):
        """positions x in [-bound, bound]^3 and unit directions d -> (sigma [...], rgb [..., 3])."""
        if not self._fused_head_ok(x):
            field = self.density(x)
            return field['sigma'], self._shade(d, field['geo_feat'])
        from .field import field_infer, infer_fusable
        from .head import rgb_head, sample_head
        if not torch.is_grad_enabled() and infer_fusable(self):
            return field_infer(self, (x + self.bound) / (2 * self.bound), d)
        sigma, colour_rows = sample_head(self._geometry_rows(x), d)
        return sigma, rgb_head(self.color_net.forward_padded(colour_rows))

    def density(self, x):
        from .field import field_fusable
        if x.is_cuda and x.dim() == 2 and torch.is_autocast_enabled() and field_fusable(self.encoder, self.sigma_net):
            h = self._geometry_rows(x)
        else:
            h = self.sigma_net(self.encoder(x, bound=self.bound))
        return {'sigma': trunc_exp(h[..., 0]), 'geo_feat': h[..., 1:]}

    def color(self, x, d, mask=None, geo_feat=None, **kwargs):
        """Colour of the samples selected by `mask` (all if None); the others get 0 — what `run()` asks for (network_ff.py:95-134)."""
        if mask is None:
            return self._shade(d, geo_feat)
        rgbs = torch.zeros(mask.shape[0], 3, dtype=x.dtype, device=x.device)
        if mask.any():
            rgbs[mask] = self._shade(d[mask], geo_feat[mask]).to(rgbs.dtype)
        return rgbs

    def run(self, rays_o, rays_d, yolo_details=None, fused=False, **kwargs):
        """`fused=True`: the fixed-step path through csrc/fixedstep.hip (same image, depth and gradients as `NeRFRenderer.run`, which
        stays the default)."""
        if fused and kwargs.get("upsample_steps", 0) == 0 and self.bg_radius <= 0:
            from .fixedstep import render_fixed_steps
            kwargs.pop("upsample_steps", None)
            return render_fixed_steps(self, rays_o, rays_d, yolo_details=yolo_details, **kwargs)
        return super().run(rays_o, rays_d, yolo_details, **kwargs)

    def get_params(self, lr):
        """One optimizer group per sub-module, as the reference trainer expects (network_ff.py:137-148)."""
        return [{'params': m.parameters(), 'lr': lr} for m in (self.encoder, self.sigma_net, self.encoder_dir, self.color_net)]
