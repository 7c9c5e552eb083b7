"""Generates tests/golden/*.npz from the parts of the reference that run on CPU.

Run ONLY in the build container (needs /root/reference); the GPU box never runs this.
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is imported from the reference (nothing is copied into this repo; the fixtures hold data only):
  * activation.trunc_exp                      -> trunc_exp.npz   (forward + backward values)
  * nerf.renderer.NeRFRenderer.run            -> run_foc.npz     (FOC fixed-step compositing, mask w > 1e-10)
  (legacy/nerf/renderer.py is not importable here: its `from .utils import custom_meshgrid` pulls in
   imageio, cv2, tensorboardX, mcubes, lpips, torchmetrics, torch_ema ... none of which are installed.)
`raymarching` (a CUDA extension that would JIT-build on import) and `trimesh`
(absent) are stubbed in sys.modules BEFORE the import; the stub's near_far_from_aabb is this
repo's CPU oracle, and its outputs are stored in the fixture as inputs of the composite.
The density / colour fields are analytic functions defined here.
"""
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
import oracle  # noqa: E402

# ---- stubs, installed before any reference import
rm = types.ModuleType("raymarching")


def _near_far(rays_o, rays_d, aabb, min_near=0.2):
    n, f = oracle.near_far_from_aabb(rays_o.detach().numpy(), rays_d.detach().numpy(), aabb.detach().numpy(), min_near)
    return torch.from_numpy(n), torch.from_numpy(f)


rm.near_far_from_aabb = _near_far
sys.modules["raymarching"] = rm
sys.modules["trimesh"] = types.ModuleType("trimesh")
sys.path.insert(0, REF)

from activation import trunc_exp  # noqa: E402
import nerf.renderer as foc_renderer  # noqa: E402


def sigma_field(x):
    c = torch.tensor([0.1, -0.05, 0.2])
    r2 = ((x - c) ** 2).sum(-1)
    return 40.0 * torch.exp(-r2 / 0.12) + 3.0 * torch.exp(-((x + 0.4) ** 2).sum(-1) / 0.02)


def color_field(x, d):
    return 0.5 + 0.5 * torch.sin(3.0 * x + 0.7 * d)


def make_rays(N, seed, bound):
    g = torch.Generator().manual_seed(seed)
    o = torch.randn(N, 3, generator=g)
    o = o / o.norm(dim=-1, keepdim=True) * (1.6 * bound)
    tgt = (torch.rand(N, 3, generator=g) - 0.5) * bound
    d = tgt - o
    d = d / d.norm(dim=-1, keepdim=True)
    # a few rays whose line misses the box entirely: tangent direction at distance 2*bound > sqrt(3)*bound
    o[::17] = o[::17] * 1.25
    t = torch.cross(o[::17], torch.tensor([[0.3, -0.5, 0.8]]).expand_as(o[::17]), dim=-1)
    d[::17] = t / t.norm(dim=-1, keepdim=True)
    return o.float(), d.float()


def run_reference(mod, foc, bound, N, T, seed):
    class Toy(mod.NeRFRenderer):
        def density(self, x):
            return {'sigma': sigma_field(x), 'geo_feat': x[..., :2] * 0.0}

        if foc:
            def color(self, x, d, yolo_details, mask=None, geo_feat=None, **kw):
                rgbs = torch.zeros(mask.shape[0], 3)
                rgbs[mask] = color_field(x[mask], d[mask])
                return rgbs
        else:
            def color(self, x, d, mask=None, geo_feat=None, **kw):
                rgbs = torch.zeros(mask.shape[0], 3)
                rgbs[mask] = color_field(x[mask], d[mask])
                return rgbs

    m = Toy(bound=bound, min_near=0.2)
    m.eval()
    o, d = make_rays(N, seed, bound)
    with torch.no_grad():
        if foc:
            res = m.run(o[None], d[None], None, num_steps=T, upsample_steps=0, bg_color=None, perturb=False)
        else:
            res = m.run(o[None], d[None], num_steps=T, upsample_steps=0, bg_color=None, perturb=False)
    nears, fars = _near_far(o, d, m.aabb_infer, m.min_near)
    # fields at the sample positions the renderer used (recomputed exactly as run() does)
    z = torch.linspace(0.0, 1.0, T)[None].expand(N, T)
    z = nears[:, None] + (fars - nears)[:, None] * z
    xyz = o[:, None, :] + d[:, None, :] * z[..., None]
    xyz = torch.min(torch.max(xyz, m.aabb_infer[:3]), m.aabb_infer[3:])
    sig = sigma_field(xyz.reshape(-1, 3)).view(N, T)
    out = dict(rays_o=o.numpy(), rays_d=d.numpy(), aabb=m.aabb_infer.numpy(), min_near=np.float32(m.min_near), T=np.int32(T),
               nears=nears.numpy(), fars=fars.numpy(), sigmas=sig.numpy(),
               image=res['image'][0].numpy(), depth=res['depth'][0].numpy(), weights_sum=res['weights_sum'].numpy())
    if foc:
        out['rgbs'] = res['rgbs'].numpy()
        assert np.array_equal(res['densities'].squeeze(-1).numpy(), sig.numpy())
    else:
        # legacy run() does not return the fields: recompute colour with its w > 1e-4 mask
        dl = z[:, 1:] - z[:, :-1]
        dl = torch.cat([dl, ((fars - nears) / T)[:, None]], -1)
        al = 1 - torch.exp(-dl * sig)
        w = al * torch.cumprod(torch.cat([torch.ones_like(al[:, :1]), 1 - al + 1e-15], -1), -1)[:, :-1]
        msk = w > 1e-4
        rgbs = torch.zeros(N, T, 3)
        dd = d[:, None, :].expand(N, T, 3)
        rgbs[msk] = color_field(xyz[msk], dd[msk])
        out['rgbs'] = rgbs.numpy()
    return out


def main():
    # trunc_exp
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(256, generator=g) * 8).requires_grad_(True)
    y = trunc_exp(x)
    gy = torch.randn(256, generator=g)
    y.backward(gy)
    np.savez_compressed(os.path.join(HERE, "trunc_exp.npz"), x=x.detach().numpy(), y=y.detach().numpy(), gy=gy.numpy(), gx=x.grad.numpy())

    np.savez_compressed(os.path.join(HERE, "run_foc.npz"), **run_reference(foc_renderer, True, bound=1, N=96, T=128, seed=1))
    np.savez_compressed(os.path.join(HERE, "run_foc_b2.npz"), **run_reference(foc_renderer, True, bound=2, N=64, T=512, seed=2))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
