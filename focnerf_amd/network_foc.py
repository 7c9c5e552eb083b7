"""FOC's object-conditioned NeRF network without tinycudann — the topology and method signatures of the reference's
nerf/network_tcnn.py:451-681 (`NeRFNetwork` with a YOLO object feature in the colour input) on this repo's GridEncoder + FFMLP
(SURVEY.md §8f-1).

    sigma-net   hash grid 32 -> 64 -> 64 -> 16                  (trunc_exp on channel 0, 15 geometry features)
    yolo_feat_encoder   144 -> 16 (ReLU) -> 16, no biases        (one vector per image; tcnn FullyFusedMLP, n_neurons 16)
    colour-net  [SH16(d) | geo 15 | encoded object feature 16 | pad 1] = 48 -> 64 -> 64 -> 16, sigmoid on 0..2

Differences from the tcnn file, all forced by the in-tree FFMLP (ffmlp.py:83-86) and recorded for parity purposes:
FFMLP has at least two hidden layers, so `num_layers=2` means 32->64->64->16 like nerf/network_ff.py:31 (tcnn: one hidden
layer); the 47-wide colour input is padded to 48 (tcnn pads to a multiple of 16 internally). tinycudann is an un-vendored,
unpinned dependency of the reference: **parity unpinned by construction**; the tests pin this file to the CPU oracle's
chain (oracle.grid_encode_forward -> ffmlp_forward -> oracle/torch_cpu_nerf.sh_encode_deg4 -> 48-wide ffmlp_forward), to its own
torch expressions and to `network.NeRFNetwork` with a zero object feature.

Fast paths (round 3): the object feature is ONE 16-vector per object, so its share of the colour network's first layer,
W0[:, 31:47] . obj, is a constant per neuron. The fused training tail (`fixedstep._render_tail`), the whole-field inference kernel
(`field.field_infer`) and the combiner's producer (`fixedstep.render_field4`) take it as the initial value of the layer-0 accumulators
and otherwise run the 32-wide kernels of `network.NeRFNetwork`; the gradient of the encoded feature (for `yolo_feat_encoder`) and of
W0[:, 31:47] follow from the column sum of the first layer's delta (csrc/ffmlp.hip, MlpHead).
"""
import os

import torch
import torch.nn as nn

from .activation import trunc_exp
from .encoding import get_encoder
from .ffmlp import FFMLP
from .renderer import NeRFRenderer


class _tiny_mlp_vec(torch.autograd.Function):
    """y = W1 relu(W0 x) for ONE input vector, fp32, as matrix-vector products: the object feature is one 144-vector per image, and as two
    `nn.Linear` under autocast it was two casts of the weights, two hipBLASLt GEMM launches of ~40 us each for 2 x 144 x 16 multiplies, and
    four more in the backward — a quarter millisecond of GPU time and a third of the step's host time for a 2.5 k-parameter encoder."""

    @staticmethod
    def forward(ctx, x, w0, w1):
        h = torch.mv(w0, x)
        a = torch.relu(h)
        ctx.save_for_backward(x, w0, w1, a)
        return torch.mv(w1, a)

    @staticmethod
    def backward(ctx, g):
        x, w0, w1, a = ctx.saved_tensors
        g = g.float()
        g_h = torch.mv(w1.t(), g) * (a > 0)
        return (torch.mv(w0.t(), g_h) if ctx.needs_input_grad[0] else None), torch.outer(g_h, x), torch.outer(g, a)


class _TinyMLP(nn.Module):
    """in -> 16 -> out, ReLU, no biases (tcnn FullyFusedMLP with one hidden layer of 16 neurons, network_tcnn.py:502-514)."""

    def __init__(self, in_dim, out_dim, hidden=16):
        super().__init__()
        self.l0 = nn.Linear(in_dim, hidden, bias=False)
        self.l1 = nn.Linear(hidden, out_dim, bias=False)

    def forward(self, x):
        if x.dim() == 2 and x.shape[0] == 1 and x.is_cuda:          # the one-vector case of the path, in fp32 whatever the autocast state
            with torch.autocast("cuda", enabled=False):
                return _tiny_mlp_vec.apply(x[0].float(), self.l0.weight.float(), self.l1.weight.float()).unsqueeze(0)
        return self.l1(torch.relu(self.l0(x)))


class NeRFNetwork(NeRFRenderer):
    uses_object_feature = True

    def __init__(self, encoding="HashGrid", encoding_dir="SphericalHarmonics", num_layers=2, hidden_dim=64, geo_feat_dim=15,
                 num_layers_color=3, hidden_dim_color=64, yolo_encoding_dim=16, bound=1, n_chunks=5, yolo_feats_encoder_dim=144, **kwargs):
        super().__init__(bound, **kwargs)
        self.num_layers, self.hidden_dim, self.geo_feat_dim = num_layers, hidden_dim, geo_feat_dim
        self.yolo_encoding_dim, self.yolo_feats_encoder_dim, self.n_chunks = yolo_encoding_dim, yolo_feats_encoder_dim, n_chunks

        self.encoder, self.in_dim = get_encoder("hashgrid", desired_resolution=2048 * bound)          # :478-488
        self.sigma_net = FFMLP(input_dim=self.in_dim, output_dim=1 + self.geo_feat_dim, hidden_dim=hidden_dim, num_layers=num_layers)
        self.yolo_feat_encoder = self.get_yolo_feat_encoder(yolo_feats_encoder_dim)                    # :502-514

        self.num_layers_color = 2                                                                     # :517 (the attribute; the net uses the argument)
        self.hidden_dim_color = 64
        self.encoder_dir, sh_dim = get_encoder("sphere_harmonics")                                    # degree 4, :520-526
        self.in_dim_color = sh_dim + self.geo_feat_dim                                                # 31, :528
        self.color_in = self.in_dim_color + self.yolo_encoding_dim                                    # 47
        self.color_in_padded = (self.color_in + 15) // 16 * 16                                        # 48
        self.color_net = FFMLP(input_dim=self.color_in_padded, output_dim=3, hidden_dim=hidden_dim_color, num_layers=num_layers_color - 1)

    # ------------------------------------------------------------------ object feature
    def get_yolo_feat_encoder(self, yolo_feats_encoder_dim):
        return _TinyMLP(yolo_feats_encoder_dim, self.yolo_encoding_dim)

    def encode_object_feature(self, yolo_details, device):
        """yolo_details = (mask, bbox, raw object feature [yolo_feats_encoder_dim]) -> [16] (network_tcnn.py:607-613)."""
        if yolo_details is None:
            return torch.zeros(self.yolo_encoding_dim, device=device)
        def encode():
            raw = torch.as_tensor(yolo_details[2], device=device, dtype=torch.float32)
            return self.yolo_feat_encoder(raw.unsqueeze(0)).squeeze(0)
        if torch.is_grad_enabled():
            return encode()
        from .field import scope_cached          # a staged render asks once per ray chunk: one encoding per half_cache_scope
        return scope_cached(("object_feature", id(self), id(yolo_details[2])), yolo_details[2], encode)

    # ------------------------------------------------------------------ field
    def _fused_ok(self, x):
        from .shencoder import SHEncoder
        return (x.is_cuda and x.dim() == 2 and torch.is_autocast_enabled() and self.geo_feat_dim == 15 and self.yolo_encoding_dim == 16
                and isinstance(self.encoder_dir, SHEncoder) and os.environ.get("FOC_FUSED_HEAD", "1") != "0")

    def _sigma_features(self, x):
        from .field import field_fusable, hashgrid_mlp
        if x.is_cuda and x.dim() == 2 and torch.is_autocast_enabled() and field_fusable(self.encoder, self.sigma_net):
            return hashgrid_mlp(self.encoder, self.sigma_net, x, self.bound)
        return self.sigma_net(self.encoder(x, bound=self.bound))

    def forward(self, x, d, yolo_details=None):
        """:555-586. The reference concatenates `yolo_details[2]` as a per-sample feature here; a [16] vector is broadcast."""
        obj = yolo_details[2] if yolo_details is not None else torch.zeros(self.yolo_encoding_dim, device=x.device)
        obj = torch.as_tensor(obj, device=x.device)
        if self._fused_ok(x) and obj.numel() == self.yolo_encoding_dim:
            from .field import field_infer, infer_fusable
            from .head import sample_head, rgb_head
            if not torch.is_grad_enabled() and infer_fusable(self):
                return field_infer(self, (x + self.bound) / (2 * self.bound), d, obj_feat=obj)
            sigma, cin = sample_head(self._sigma_features(x), d, obj)
            return sigma, rgb_head(self.color_net.forward_padded(cin))
        h = self._sigma_features(x)
        sigma = trunc_exp(h[..., 0])
        geo_feat = h[..., 1:]
        return sigma, self._color_torch(d, geo_feat, obj)

    def density(self, x, yolo_details=None):
        h = self._sigma_features(x)
        return {'sigma': trunc_exp(h[..., 0]), 'geo_feat': h[..., 1:]}

    def _color_torch(self, d, geo_feat, obj_feat):
        d = self.encoder_dir(d)
        obj = obj_feat.to(geo_feat.dtype)
        if obj.dim() == 1:
            obj = obj.unsqueeze(0).expand(geo_feat.shape[0], -1)
        pad = torch.zeros_like(geo_feat[..., :self.color_in_padded - self.color_in])
        h = torch.cat([d.to(geo_feat.dtype), geo_feat, obj, pad], dim=-1)
        return torch.sigmoid(self.color_net(h))

    def color(self, x, d, yolo_details=None, mask=None, geo_feat=None, **kwargs):
        """Colour of the samples selected by `mask` (all if None), conditioned on the encoded object feature; the others get 0
        (network_tcnn.py:607-654)."""
        obj_feat = self.encode_object_feature(yolo_details, x.device)
        if mask is None:
            return self._color_torch(d, geo_feat, obj_feat)
        rgbs = torch.zeros(mask.shape[0], 3, dtype=x.dtype, device=x.device)
        if mask.any():
            rgbs[mask] = self._color_torch(d[mask], geo_feat[mask], obj_feat).to(rgbs.dtype)
        return rgbs

    def run(self, rays_o, rays_d, yolo_details=None, fused=False, **kwargs):
        if fused and kwargs.get("upsample_steps", 0) == 0 and self.bg_radius <= 0:
            from .fixedstep import render_fixed_steps
            kwargs.pop("upsample_steps", None)
            return render_fixed_steps(self, rays_o, rays_d, yolo_details=yolo_details, **kwargs)
        return super().run(rays_o, rays_d, yolo_details, **kwargs)

    def get_params(self, lr):
        """One optimizer group per sub-module, the object-feature encoder included."""
        parts = (self.encoder, self.sigma_net, self.encoder_dir, self.color_net, self.yolo_feat_encoder)
        return [{'params': m.parameters(), 'lr': lr} for m in parts]
