"""Hierarchical resampling of the fixed-step renderer (upsample_steps > 0): `sample_pdf` (nerf/renderer.py:13-46) and the legacy
renderer's run() (legacy/nerf/renderer.py:125-254) against tests/golden/upsample.npz, which the reference's own code produced
(make_golden.py upsample_fixture). CPU: sample_pdf bit for bit. GPU: NeRFRenderer.run on the HIP near/far op with an analytic field."""
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _have_lib():
    return os.path.exists(os.path.join(os.path.dirname(GOLDEN), "..", "focnerf_amd", "libfocnerf_hip.so"))


def test_sample_pdf_reproduces_the_reference_bit_for_bit():
    from focnerf_amd.renderer import sample_pdf
    g = np.load(os.path.join(GOLDEN, "upsample.npz"))
    bins, w = torch.from_numpy(g["pdf_bins"]), torch.from_numpy(g["pdf_weights"])
    got = sample_pdf(bins, w, 16, det=True)
    assert np.array_equal(got.numpy(), g["pdf_samples_det"])
    # samples lie inside the bins' range and are sorted along a row (strata centres through a monotone inverse CDF)
    assert torch.all(got >= bins[:, :1] - 1e-6) and torch.all(got <= bins[:, -1:] + 1e-6) and torch.all(got[:, 1:] >= got[:, :-1])
    torch.manual_seed(0)
    rnd = sample_pdf(bins, w, 64, det=False)
    assert rnd.shape == (25, 64) and torch.all(rnd >= bins[:, :1] - 1e-6) and torch.all(rnd <= bins[:, -1:] + 1e-6)
    # where the weight is concentrated the samples are too: row 4 has no mass in its first ten intervals
    assert (rnd[4] < bins[4, 9]).float().mean() < 0.02


@pytest.mark.gpu
def test_run_with_upsample_steps_matches_the_legacy_renderer():
    """legacy/nerf/renderer.py run(num_steps=48, upsample_steps=32) on an analytic field -> image, depth, weights_sum; the same field
    through focnerf_amd's NeRFRenderer.run on the device (the legacy colour mask w > 1e-4 is `weight_thresh`)."""
    from focnerf_amd.renderer import NeRFRenderer
    g = np.load(os.path.join(GOLDEN, "upsample.npz"))

    class Toy(NeRFRenderer):
        def density(self, x):
            c = torch.tensor([0.1, -0.05, 0.2], device=x.device)
            r2 = ((x - c) ** 2).sum(-1)
            sig = 40.0 * torch.exp(-r2 / 0.12) + 3.0 * torch.exp(-((x + 0.4) ** 2).sum(-1) / 0.02)
            return {'sigma': sig, 'geo_feat': x[..., :2] * 0.5}

        def color(self, x, d, mask=None, geo_feat=None, **kw):
            rgbs = torch.zeros(mask.shape[0], 3, device=x.device)
            col = 0.5 + 0.5 * torch.sin(3.0 * x[mask] + 0.7 * d[mask])
            rgbs[mask] = col * (0.5 + geo_feat[mask][..., :1].abs().clamp(max=0.5))
            return rgbs

    m = Toy(bound=1).cuda().eval()
    o, d = torch.from_numpy(g["rays_o"]).cuda(), torch.from_numpy(g["rays_d"]).cuda()
    T, t = int(g["T"]), int(g["t"])
    with torch.no_grad():
        out = m.run(o[None], d[None], num_steps=T, upsample_steps=t, perturb=False, weight_thresh=1e-4, return_fields=True)
        coarse = m.run(o[None], d[None], num_steps=T, upsample_steps=0, perturb=False, weight_thresh=1e-4)
    hit = g["nears"] < 1e30
    assert out["densities"].shape == (o.shape[0], T + t, 1) and out["rgbs"].shape == (o.shape[0], T + t, 3)
    np.testing.assert_allclose(out["image"][0].cpu().numpy(), g["image"], atol=2e-4, rtol=0)
    np.testing.assert_allclose(out["weights_sum"].cpu().numpy(), g["weights_sum"], atol=2e-4, rtol=0)
    np.testing.assert_allclose(out["depth"][0].cpu().numpy()[hit], g["depth"][hit], atol=2e-4, rtol=0)
    np.testing.assert_allclose(coarse["image"][0].cpu().numpy(), g["image_coarse"], atol=2e-4, rtol=0)
    assert np.abs(g["image"] - g["image_coarse"]).max() > 1e-3          # the resampling does change the picture
    # training mode draws random quantiles: still a valid render close to the deterministic one
    m.train()
    torch.manual_seed(0)
    tr = m.run(o[None], d[None], num_steps=T, upsample_steps=t, perturb=True, weight_thresh=1e-4)
    assert torch.isfinite(tr["image"][0][torch.from_numpy(hit).cuda()]).all()
    assert (tr["image"][0].cpu().numpy()[hit] - g["image"][hit]).__abs__().max() < 0.1
