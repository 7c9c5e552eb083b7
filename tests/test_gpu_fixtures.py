"""GPU: the fixtures generated from the IMPORTED reference (tests/golden/make_golden.py) replayed directly on the device kernels —
round 1 reached them only through the oracle (HIP <-> oracle <-> fixture). run_foc*.npz = nerf.renderer.NeRFRenderer.run (FOC's
fixed-step renderer, nerf/renderer.py:126-238) on an analytic field; trunc_exp.npz = activation.trunc_exp forward / backward."""
import os

import numpy as np
import pytest
import torch

import oracle
from util import to_np

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def blocked_rows_of(N, T, block=64):
    """Ray-major row of every row of the block-interleaved order (include/focnerf.h: row (n/64)*64*T + i*64 + n%64, padding = ray N-1)."""
    nb = -(-N // block)
    n = torch.arange(nb * block).clamp(max=N - 1).view(nb, 1, block)
    return (n * T + torch.arange(T).view(1, T, 1)).reshape(-1)


@pytest.mark.parametrize("name", ["run_foc.npz", "run_foc_b2.npz"])
def test_reference_run_fixture_through_the_inference_kernel(name):
    """foc_fixed_render_inference (k_fs_render_infer: weights by wave scan, mask w > 1e-10, composite, depth) on the sigma / rgb the
    reference's run() saw: image, depth and weights_sum within the 1e-4 target, NaN depth on rays that miss the box (0 * NaN, as in the
    reference), the returned `rgbs` field equal to the reference's. Also the near/far kernel on the fixture's rays, bit for bit."""
    from focnerf_amd._lib import lib, ptr, stream_of, check
    from focnerf_amd import raymarching
    g = np.load(os.path.join(GOLDEN, name))
    N, T = g["sigmas"].shape
    nears, fars = raymarching.near_far_from_aabb(_cuda(g["rays_o"]), _cuda(g["rays_d"]), _cuda(g["aabb"]), float(g["min_near"]))
    assert np.array_equal(to_np(nears), g["nears"]) and np.array_equal(to_np(fars), g["fars"])
    sigma, rgb = _cuda(g["sigmas"]).reshape(-1), _cuda(g["rgbs"]).reshape(-1, 3)
    image, depth, ws = torch.empty(N, 3, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    masked = torch.empty(N * T, 3, device="cuda")
    check(lib.foc_fixed_render_inference(ptr(sigma), ptr(rgb), ptr(nears), ptr(fars), None, None, 1.0, N, T, 1.0, 1e-10, ptr(image), ptr(depth), ptr(ws),
                                         ptr(masked), 0, None, stream_of(sigma)), "fixed_render_inference")
    hit = g["nears"] < 1e30
    assert hit.sum() > 0.7 * N and (~hit).sum() > 0
    np.testing.assert_allclose(to_np(image), g["image"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(to_np(ws), g["weights_sum"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(to_np(depth)[hit], g["depth"][hit], rtol=0, atol=1e-4)
    assert np.isnan(to_np(depth)[~hit]).all() and np.isnan(g["depth"][~hit]).all()
    # the fixture's rgbs are already masked by the reference's w > 1e-10: masking them again changes nothing, except where the decision
    # hangs on the last bit of exp(): for delta * sigma ~ 3e-8 torch's CPU exp returns 1 - 2^-24 (alpha = 6e-8, weight > 1e-10, colour
    # kept) where the correctly rounded value — libm in the oracle, the device here — is 1 (alpha = 0, weight 0). Such a sample carries
    # at most 6e-8 of the image either way.
    same = (to_np(masked).reshape(N, T, 3) == g["rgbs"]).all(-1)
    step = (g["fars"] - g["nears"])[:, None] / (T - 1)
    x = step * g["sigmas"]                                        # delta * sigma, to well within a factor of 2
    assert same.mean() > 0.98 and np.all(x[~same] < 2.0 ** -22), (same.mean(), x[~same].max() if (~same).any() else 0)
    # the packed producer is the same kernel: identical outputs, field4 = (sigma, masked rgb)
    f4 = torch.empty(N, T, 4, device="cuda")
    image2, depth2, ws2 = torch.empty_like(image), torch.empty_like(depth), torch.empty_like(ws)
    check(lib.foc_fixed_field_pack(ptr(sigma), ptr(rgb), ptr(nears), ptr(fars), None, None, 1.0, N, T, 1.0, 1e-10, ptr(image2), ptr(depth2), ptr(ws2),
                                   ptr(f4), 0, stream_of(sigma)), "fixed_field_pack")
    assert torch.equal(image, image2) and torch.equal(ws, ws2) and torch.equal(torch.nan_to_num(depth), torch.nan_to_num(depth2))
    assert torch.equal(f4[..., 0].reshape(-1), sigma) and torch.equal(f4[..., 1:].reshape(-1, 3), masked)
    # the block-interleaved sample order of the render path (64 rays x T depths per block, last block padded): same kernel body per ray,
    # staged through LDS — identical bits, per-sample outputs ray-major again
    rows = blocked_rows_of(N, T).cuda()
    sig_b, rgb_b = sigma[rows].contiguous(), rgb[rows].contiguous()
    image3, depth3, ws3, masked3, sig_rm = torch.empty_like(image), torch.empty_like(depth), torch.empty_like(ws), torch.empty_like(masked), torch.empty_like(sigma)
    check(lib.foc_fixed_render_inference(ptr(sig_b), ptr(rgb_b), ptr(nears), ptr(fars), None, None, 1.0, N, T, 1.0, 1e-10, ptr(image3), ptr(depth3), ptr(ws3),
                                         ptr(masked3), 64, ptr(sig_rm), stream_of(sigma)), "fixed_render_inference")
    assert torch.equal(image, image3) and torch.equal(ws, ws3) and torch.equal(torch.nan_to_num(depth), torch.nan_to_num(depth3))
    assert torch.equal(masked, masked3) and torch.equal(sig_rm, sigma)
    f4b = torch.empty_like(f4)
    check(lib.foc_fixed_field_pack(ptr(sig_b), ptr(rgb_b), ptr(nears), ptr(fars), None, None, 1.0, N, T, 1.0, 1e-10, None, None, None, ptr(f4b), 64,
                                   stream_of(sigma)), "fixed_field_pack")
    assert torch.equal(f4, f4b)
    # and the combiner's composite (image_depth_generation is a copy of run()'s compositing, COMBINED.py:141-200)
    from focnerf_amd.combine import composite_fixed_steps
    img4, dep4 = composite_fixed_steps(_cuda(g["sigmas"]), _cuda(g["rgbs"]), nears, fars, 1.0)
    np.testing.assert_allclose(to_np(img4)[:, :3], np.clip(g["image"], 0, 1), rtol=0, atol=1e-4)
    np.testing.assert_allclose(to_np(dep4)[hit], g["depth"][hit], rtol=0, atol=1e-4)


@pytest.mark.parametrize("name", ["run_foc.npz", "run_foc_b2.npz"])
def test_reference_run_fixture_through_the_training_tail(name):
    """foc_fixed_tail_forward (k_fs_tail_fwd: trunc_exp of the density logit, sigmoid of the colour logits, weights, mask, composite —
    what the training step runs) on the fixture. The tail takes fp16 LOGITS, so sigma and rgb enter as half(log sigma) and
    half(logit rgb): against the reference's numbers the tolerance is that quantisation (relative 2^-11 * |log sigma| on sigma:
    image / weights within 5e-3), against the fixture-pinned CPU oracle evaluated on the SAME quantised values the 1e-4 target."""
    from focnerf_amd._lib import lib, ptr, stream_of, check
    g = np.load(os.path.join(GOLDEN, name))
    N, T = g["sigmas"].shape
    M = N * T
    sig, rgb = g["sigmas"].astype(np.float64), g["rgbs"].astype(np.float64)
    h0 = np.log(np.maximum(sig, 1e-30)).clip(-60, 60).astype(np.float16)                  # sigma = 0 -> exp(-60) ~ 1e-26
    kept = (g["rgbs"] != 0).any(-1)                                                       # the reference queried colour here (w > 1e-10)
    logit = np.log(np.clip(rgb, 1e-6, 1 - 1e-6) / (1 - np.clip(rgb, 1e-6, 1 - 1e-6))).astype(np.float16)
    h = torch.zeros(M, 16, dtype=torch.float16, device="cuda")
    h[:, 0] = _cuda(h0.reshape(-1))
    c = torch.zeros(M, 4, dtype=torch.float16, device="cuda")
    c[:, :3] = _cuda(logit.reshape(-1, 3))
    nears, fars = _cuda(g["nears"]), _cuda(g["fars"])
    sigma, trans, weights = (torch.empty(M, device="cuda") for _ in range(3))
    ws, depth, image = torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, 3, device="cuda")
    check(lib.foc_fixed_tail_forward(ptr(h), ptr(c), ptr(nears), ptr(fars), None, None, 1.0, N, T, 1.0, 1e-10, ptr(sigma), ptr(trans), ptr(weights), ptr(ws),
                                     ptr(depth), ptr(image), 4, None, stream_of(h)), "fixed_tail_forward")
    hit = g["nears"] < 1e30
    # (1) against the reference's own outputs, tolerance = fp16 quantisation of the logits
    np.testing.assert_allclose(to_np(ws), g["weights_sum"], rtol=0, atol=5e-3)
    np.testing.assert_allclose(to_np(image)[hit], g["image"][hit], rtol=0, atol=5e-3)
    np.testing.assert_allclose(to_np(depth)[hit], g["depth"][hit], rtol=0, atol=5e-3)
    # (2) against the oracle on the quantised inputs (sigma' = exp(h0) in fp32, rgb' = half(sigmoid(logit)) where the reference kept the
    # sample): 1e-4. The oracle's composite is pinned to this very fixture on the CPU (tests/test_oracle.py).
    sig_q = np.exp(h0.astype(np.float32))
    np.testing.assert_allclose(to_np(sigma).reshape(N, T), sig_q, rtol=2e-6, atol=0)
    rgb_q = (1.0 / (1.0 + np.exp(-logit.astype(np.float32)))).astype(np.float16).astype(np.float32)
    i4, d_o, w_o = oracle.composite_fixed_steps(sig_q, rgb_q, g["nears"], g["fars"], bg=1.0, clamp01=False, want_weights=True)
    np.testing.assert_allclose(to_np(weights).reshape(N, T), w_o, rtol=0, atol=1e-4)
    np.testing.assert_allclose(to_np(ws), w_o.sum(-1), rtol=0, atol=1e-4)
    np.testing.assert_allclose(to_np(image)[hit], i4[hit, :3], rtol=0, atol=1e-4)
    np.testing.assert_allclose(to_np(depth)[hit], d_o[hit], rtol=0, atol=1e-4)
    assert kept.any()


def test_reference_trunc_exp_fixture_through_the_head_kernels():
    """activation.trunc_exp's own forward / backward values (trunc_exp.npz: x ~ 8 * N(0,1), so the +-15 clamp of the backward is hit)
    through k_head_fwd / k_head_bwd (focnerf_amd.head.sample_head: sigma = exp(h[:,0]), grad_h[:,0] = g_sigma * exp(clamp(h0, -15, 15))).
    The kernels read the logit as fp16 — that is the data type of the sigma network's output — so x enters rounded to half:
    tolerance vs the fixture = that rounding (relative |x| * 2^-11 on y; same on gx, plus the half rounding of grad_h), and vs torch's
    trunc_exp on the rounded x (bit-identical to the reference on the CPU, tests/test_oracle.py) 2 fp32 ulps / 1 half ulp."""
    from focnerf_amd.activation import trunc_exp
    from focnerf_amd.head import sample_head
    g = np.load(os.path.join(GOLDEN, "trunc_exp.npz"))
    x, y, gy, gx = g["x"], g["y"], g["gy"], g["gx"]
    assert (np.abs(x) > 15).sum() > 5
    M = x.shape[0]
    xh = x.astype(np.float16)
    h = torch.zeros(M, 16, dtype=torch.float16, device="cuda")
    h[:, 0] = _cuda(xh)
    h.requires_grad_(True)
    dirs = torch.nn.functional.normalize(torch.randn(M, 3, device="cuda"), dim=-1)
    sigma, cin = sample_head(h, dirs)
    scale = 2.0 ** -14                                           # exact power of two: keeps gy * exp(15) ~ 1e7 inside fp16's range
    sigma.backward(_cuda(gy) * scale)
    got_y, got_gx = to_np(sigma), to_np(h.grad[:, 0].float()) / scale
    # vs the fixture, through the half rounding of x
    rel = np.abs(x) * 2.0 ** -11 + 1e-6
    fin = np.isfinite(y)
    assert np.all(np.abs(got_y[fin] - y[fin]) <= rel[fin] * np.abs(y[fin]) + 1e-30)
    tol_gx = (np.where(np.abs(x) < 15, rel, 1e-6) + 2.0 ** -10) * np.abs(gx) + 2.0 ** -24 / scale
    assert np.all(np.abs(got_gx - gx) <= tol_gx), np.abs(got_gx - gx).max()
    # vs the reference-pinned torch op on the rounded logit
    xt = torch.from_numpy(xh.astype(np.float32)).requires_grad_(True)
    yt = trunc_exp(xt)
    yt.backward(torch.from_numpy(gy) * scale)
    np.testing.assert_allclose(got_y, yt.detach().numpy(), rtol=3e-7, atol=0)
    want_g = xt.grad.numpy()
    assert np.all(np.abs(to_np(h.grad[:, 0].float()) - want_g) <= 2.0 ** -10 * np.abs(want_g) + 2.0 ** -24)
    # the clamp is really in play: |x| > 15 entries have gx = gy * exp(+-15)
    big = np.abs(x) > 15.5
    np.testing.assert_allclose(got_gx[big], gy[big] * np.exp(np.sign(x[big]) * 15.0), rtol=2e-3, atol=2.0 ** -24 / scale)   # half underflow of the scaled gradient
