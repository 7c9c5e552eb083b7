"""GPU: the ops driven end-to-end through the renderer / network (the callers of SURVEY.md §3a-3c)."""
import numpy as np
import pytest
import torch

import oracle
from util import to_np

pytestmark = pytest.mark.gpu


def _model(bound, cuda_ray, seed=0):
    from focnerf_amd import synthetic
    from focnerf_amd.network import NeRFNetwork
    torch.manual_seed(seed)
    m = NeRFNetwork(bound=bound, cuda_ray=cuda_ray).cuda()
    m.encoder.embeddings.data.uniform_(-0.5, 0.5)
    if cuda_ray:
        m.set_density_grid(synthetic.analytic_density_grid(bound, device="cuda"))
    return m


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()


def test_network_matches_oracle_pipeline():
    """encoder -> sigma MLP -> trunc_exp / colour MLP -> sigmoid, against the oracle ops chained on the CPU (fp32 accumulation, fp16 layer
    outputs, the ORACLE's degree-4 SH — oracle/torch_cpu_nerf.py, pinned by the reference-generated cpu_network.npz), in half-ulps: the
    kernels differ from that chain only in the order of their fp32 sums, so almost every value is the oracle's bits and the rest are one
    or two half-ulps away (a flipped rounding of a hidden activation moves the next layer's sum by ~1e-5)."""
    from oracle import torch_cpu_nerf
    from util import assert_half_close
    m = _model(1, False).eval()
    B = 1000
    x = torch.rand(B, 3, device="cuda") * 2 - 1
    d = torch.randn(B, 3, device="cuda")
    d = d / d.norm(dim=-1, keepdim=True)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        sigma, rgb = m(x, d)
    S = float(np.log2(m.encoder.per_level_scale))
    enc = oracle.grid_encode_forward(to_np((x + 1) / 2), to_np(m.encoder.embeddings).astype(np.float16), to_np(m.encoder.offsets), 3, 2, 16, S, 16)
    enc = np.transpose(enc, (1, 0, 2)).reshape(B, 32)
    pad = 128 - B % 128
    encp = np.concatenate([enc, np.zeros((pad, 32), np.float16)])
    h = oracle.ffmlp_forward(encp, to_np(m.sigma_net.weights).astype(np.float16), 32, 64, 2, 0, training=False)[:B]
    # sigma = exp(h0) in fp32: its logarithm is the density logit to ~1e-6, compared in half-ulps of the logit
    h0 = np.log(to_np(sigma))
    assert_half_close(h0, h[:, 0], ulps=2.0, atol=1e-4, what="density logit")
    same = np.abs(h0 - h[:, 0].astype(np.float32)) <= 3e-6 * np.maximum(1, np.abs(h[:, 0].astype(np.float32)))
    assert same.mean() > 0.97, f"only {same.mean():.3f} of the density logits are the oracle's bits"
    sh = torch_cpu_nerf.sh_encode_deg4(d.cpu().float()).numpy().astype(np.float16)
    cin = np.concatenate([sh, h[:, 1:], np.zeros((B, 1), np.float16)], 1)
    cin = np.concatenate([cin, np.zeros((pad, 32), np.float16)])
    c = oracle.ffmlp_forward(cin, to_np(m.color_net.weights).astype(np.float16), 32, 64, 3, 0, training=False)[:B, :3]
    rgb_ref = (1 / (1 + np.exp(-c.astype(np.float32)))).astype(np.float16).astype(np.float32)      # the half sigmoid of network_ff.py:73
    got = to_np(rgb).astype(np.float32)
    assert_half_close(got, rgb_ref, ulps=2.0, atol=1e-6, what="rgb")
    assert (got == rgb_ref).mean() > 0.95, f"only {(got == rgb_ref).mean():.3f} of the colours are the oracle's bits"


def literal_distance(B=131072, seed=1, weight_scale=1.0):
    """Distance of the HIP network's (sigma, rgb) on the BASELINE field — hash grid L16 / C2 / 2^19 with a seed-`seed` U(-1, 1) table, sigma network
    32 -> 64 -> 64 -> 16, colour network 32 -> 64 -> 64 -> 64 -> 16 with FFMLP's seed-42 initialisation (ffmlp.py:141-144) — to the oracle chain in the
    REFERENCE-LITERAL numerics: the encoder's corner sum kept in half (gridencoder.cu:164,187; oracle grid acc_mode 0) and the MLPs' running sums
    rounded to half after every 16-wide k chunk (the WMMA half accumulators of ffmlp.cu:68,169,256,458; oracle ffmlp acc_mode 1), through
    nerf/network_ff.py:51-134's glue (trunc_exp in fp32 on the half logit, half sigmoid). Also against the oracle chain in THIS library's numerics
    (fp32 accumulation, one rounding per layer). Returns a dict of plain floats."""
    from oracle import torch_cpu_nerf
    from focnerf_amd.network import NeRFNetwork
    torch.manual_seed(0)
    m = NeRFNetwork(bound=1, cuda_ray=False).cuda().eval()
    if weight_scale != 1.0:                               # a field with larger logits than the initialisation gives (a trained field's range)
        m.sigma_net.weights.data.mul_(weight_scale ** (1 / 3))
        m.color_net.weights.data.mul_(weight_scale ** (1 / 4))
    g = torch.Generator().manual_seed(seed)
    m.encoder.embeddings.data.copy_((torch.rand(m.encoder.embeddings.shape, generator=g) * 2 - 1).cuda())
    g2 = torch.Generator().manual_seed(seed + 1000)
    x = (torch.rand(B, 3, generator=g2) * 2 - 1).cuda()
    d = torch.randn(B, 3, generator=g2)
    d = (d / d.norm(dim=-1, keepdim=True)).cuda()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        sigma, rgb = m(x, d)
    sigma, rgb = to_np(sigma).astype(np.float64), to_np(rgb).astype(np.float64)
    S = float(np.log2(m.encoder.per_level_scale))
    table = to_np(m.encoder.embeddings).astype(np.float16)
    sh = torch_cpu_nerf.sh_encode_deg4(d.cpu().float()).numpy().astype(np.float16)
    Ws, Wc = to_np(m.sigma_net.weights).astype(np.float16), to_np(m.color_net.weights).astype(np.float16)
    out = {"samples": int(B), "weight_scale": float(weight_scale)}
    for name, grid_acc, mlp_acc in (("literal", 0, 1), ("fp32acc", 1, 0)):
        enc = oracle.grid_encode_forward(to_np((x + 1) / 2), table, to_np(m.encoder.offsets), 3, 2, 16, S, 16, acc_mode=grid_acc)
        enc = np.ascontiguousarray(np.transpose(enc, (1, 0, 2)).reshape(B, 32))
        h = oracle.ffmlp_forward(enc, Ws, 32, 64, 2, 0, training=False, acc_mode=mlp_acc)
        cin = np.concatenate([sh, h[:, 1:], np.zeros((B, 1), np.float16)], 1)
        c = oracle.ffmlp_forward(cin, Wc, 32, 64, 3, 0, training=False, acc_mode=mlp_acc)[:, :3]
        sig_ref = np.exp(h[:, 0].astype(np.float32)).astype(np.float64)
        rgb_ref = (1 / (1 + np.exp(-c.astype(np.float32)))).astype(np.float16).astype(np.float64)
        drgb = np.abs(rgb - rgb_ref).max(axis=1)
        dsig = np.abs(sigma - sig_ref) / sig_ref
        out[name] = {"rgb_abs_max": float(drgb.max()), "rgb_abs_p999": float(np.quantile(drgb, 0.999)), "rgb_abs_mean": float(drgb.mean()),
                     "sigma_rel_max": float(dsig.max()), "sigma_rel_p999": float(np.quantile(dsig, 0.999)), "sigma_rel_mean": float(dsig.mean()),
                     "rgb_identical_share": float((drgb == 0).mean()), "sigma_identical_share": float((dsig == 0).mean()),
                     "sigma_range": [float(sig_ref.min()), float(sig_ref.max())], "h0_abs_max": float(np.abs(h[:, 0].astype(np.float32)).max()),
                     "rgb_logit_abs_max": float(np.abs(c.astype(np.float32)).max())}
    return out


# measured on MI355X (tools/measure_literal_distance.py, gpurun_out/r5d_literal.json), bounds = measured x ~1.5:
#   initialisation-scale weights (|h0| <= 0.43, rgb logits <= 0.24), 131 072 samples: |dRGB| max / p99.9 4.88e-4 / 4.88e-4 (ONE fp16 ulp of a value in
#   [0.5, 1)), |dsigma| / sigma max / p99.9 4.89e-4 / 3.66e-4; 89.8 % of the colours are the literal model's bits;
#   weights x 16 (|h0| <= 6.2, rgb logits <= 27.5: a trained field's range), 32 768 samples: |dRGB| 5.37e-3 / 3.42e-3, |dsigma| / sigma 7.84e-3 / 5.84e-3
#   (sigma = exp(h0): an ulp of a logit in [4, 8) is 3.9e-3 relative).
LITERAL_CASES = {
    1.0: (131072, {"rgb_abs_max": 9.8e-4, "rgb_abs_p999": 7.4e-4, "sigma_rel_max": 7.5e-4, "sigma_rel_p999": 5.5e-4},
          {"rgb_identical_share": 0.95, "rgb_abs_max": 2 * 4.9e-4 + 1e-6, "sigma_rel_p999": 1.5e-4}),
    16.0: (32768, {"rgb_abs_max": 8.0e-3, "rgb_abs_p999": 5.0e-3, "sigma_rel_max": 1.2e-2, "sigma_rel_p999": 8.5e-3},
           {"rgb_identical_share": 0.95, "rgb_abs_max": 4.0e-3, "sigma_rel_p999": 1.5e-3}),
}


@pytest.mark.parametrize("weight_scale", [1.0, 16.0])
def test_end_to_end_distance_to_reference_literal_numerics(weight_scale):
    """north_star asks for 1e-4 on RGB / sigma against the reference CUDA path. The reference accumulates in HALF (encoder corner sums, WMMA
    fragments); this library accumulates in fp32 and rounds each layer once, so the network outputs differ from a literal model of the reference by
    fp16 rounding noise — one ulp of an rgb value near 0.5 is already 4.9e-4: the 1e-4 holds for the fp32 stages (composites given the same sigma /
    rgb, sample positions, the encoder bit for bit against its own numerics) and NOT end to end. This test states the end-to-end distance as numbers
    and pins it: max and 99.9th percentile of |dRGB| and |dsigma| / sigma over random samples of the BASELINE field — at the initialisation's weight
    scale and with the weights scaled up to a trained field's logit range — against measured-plus-margin bounds; against the oracle chain in this
    library's OWN numerics the outputs are the oracle's bits almost everywhere."""
    B, bounds, own_bounds = LITERAL_CASES[weight_scale]
    r = literal_distance(B, weight_scale=weight_scale)
    lit, own = r["literal"], r["fp32acc"]
    for k, bound in bounds.items():
        assert lit[k] <= bound, f"{k}: {lit[k]:.3e} above the pinned bound {bound:.1e}"
        assert lit[k] > 0.05 * bound, f"{k}: {lit[k]:.3e} — the bound {bound:.1e} is stale (more than 20x too wide), re-measure"
    # this library's own numerics model: identical bits nearly everywhere, the rest one fp16 rounding of a logit away
    assert own["rgb_identical_share"] > own_bounds["rgb_identical_share"] and own["rgb_abs_max"] <= own_bounds["rgb_abs_max"]
    assert own["sigma_rel_p999"] <= own_bounds["sigma_rel_p999"]


def test_cuda_ray_training_reduces_loss():
    """A few Adam steps on the occupancy-grid path (march -> encode -> MLPs -> composite -> backward)."""
    from focnerf_amd import synthetic
    bound = 2
    m = _model(bound, True).train()
    opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    scaler = torch.amp.GradScaler("cuda")
    o, d = synthetic.make_view_rays(64, 64, bound, 1, seed=1, device="cuda")
    target = torch.zeros(1, o.shape[1], 3, device="cuda")
    target[..., 0] = 0.8
    losses = []
    for it in range(16):
        with torch.autocast("cuda", dtype=torch.float16):
            out = m.render(o, d, staged=False, perturb=True, force_all_rays=False, dt_gamma=1 / 128, max_steps=1024, bg_color=0.0)
            loss = torch.nn.functional.mse_loss(out["image"], target)
        opt.zero_grad()
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        losses.append(loss.item())
    assert np.isfinite(losses).all()
    assert np.mean(losses[-3:]) < 0.7 * np.mean(losses[:3]), f"loss did not go down: {losses[:3]} -> {losses[-3:]}"
    # density-grid maintenance on top of the ops (morton3D, density query, packbits), inside autocast as the reference trainer runs it
    with torch.autocast("cuda", dtype=torch.float16):
        m.update_extra_state()
    assert m.mean_count > 0 and m.iter_density == 1 and m.local_step == 0
    assert m.density_bitfield.dtype == torch.uint8 and torch.isfinite(m.density_grid).all()


def test_inference_loop_matches_training_composite(monkeypatch):
    """The incremental inference path (march_rays / composite_rays) and the one-shot training path
    (march_rays_train / composite_rays_train) render the same image for the same network; and the three forms of the inference loop — the
    reference's boolean-mask compaction with a host round trip per iteration, device compaction with the count read late (Python loop), and
    one native call per iteration (csrc/occrender.hip: two-phase march, encode, whole-field kernel, composite, compaction) — give the SAME
    BITS: every ray sees the same samples in the same order whatever the burst schedule."""
    from focnerf_amd import synthetic
    bound = 2
    m = _model(bound, True, seed=3)
    o, d = synthetic.make_view_rays(48, 48, bound, 1, seed=2, device="cuda")
    kw = dict(staged=False, perturb=False, dt_gamma=1 / 128, max_steps=1024, bg_color=1.0, T_thresh=1e-4)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        m.train()
        a = m.render(o, d, staged=False, perturb=False, force_all_rays=True, dt_gamma=1 / 128, max_steps=1024, bg_color=1.0)
        m.eval()
        b = m.render(o, d, device_compaction=False, **kw)                 # the reference's loop
        assert m._native_loop_ok(o.view(-1, 3))
        dflt = m.render(o, d, **kw)                                       # default: the native loop where it serves the network
        c = m.render(o, d, device_compaction=True, **kw)                  # native loop
        monkeypatch.setenv("FOC_RENDER_NATIVE", "0")
        e = m.render(o, d, device_compaction=True, **kw)                  # Python loop, late count
        monkeypatch.setenv("FOC_RENDER_COUNT_LAG", "0")
        f = m.render(o, d, device_compaction=True, **kw)                  # Python loop, count read every iteration
    assert torch.allclose(a["image"], b["image"], atol=2e-3)
    for other in (c, e, f, dflt):
        assert torch.equal(b["image"], other["image"]) and torch.equal(b["depth"], other["depth"])
    assert (a["image"] < 0.99).any(), "the view should hit the object"


def test_fixed_step_run_matches_oracle_composite():
    """FOC default path: run() with num_steps=512; image/depth against the oracle's fixed-step composite of the same fields."""
    m = _model(1, False, seed=5).eval()
    from focnerf_amd import synthetic, raymarching
    o, d = synthetic.make_view_rays(32, 32, 1, 1, seed=3, device="cuda", radius=2.0)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        res = m.run(o, d, num_steps=512, upsample_steps=0, bg_color=1.0, perturb=False, return_fields=True)
    nears, fars = raymarching.near_far_from_aabb(o[0], d[0], m.aabb_infer, m.min_near)
    img4, depth = oracle.composite_fixed_steps(to_np(res["densities"].squeeze(-1).float()), to_np(res["rgbs"].float()), to_np(nears), to_np(fars), 1.0, clamp01=False)
    hit = to_np(nears) < 1e30
    np.testing.assert_allclose(to_np(res["image"][0])[hit], img4[hit, :3], atol=1e-4)
    np.testing.assert_allclose(to_np(res["depth"][0])[hit], depth[hit], atol=1e-4)
    # the HIP fixed-step composite used by the combiner gives the same numbers (clamped)
    from focnerf_amd.combine import composite_fixed_steps
    im4, dp = composite_fixed_steps(res["densities"].squeeze(-1).float().contiguous(), res["rgbs"].float().contiguous(), nears, fars, 1.0)
    np.testing.assert_allclose(to_np(im4)[hit], np.clip(img4[hit], 0, 1), atol=1e-4)
    np.testing.assert_allclose(to_np(dp)[hit], depth[hit], atol=1e-4)


def test_fused_head_matches_torch_glue(monkeypatch):
    """NeRFNetwork.forward through csrc/head.hip (sample_head / rgb_head) against the torch expressions of
    nerf/network_ff.py:51-75 that it replaces: same sigma and rgb, same parameter gradients."""
    m = _model(1, False).train()
    B = 5000 + 37
    gen = torch.Generator(device="cuda").manual_seed(3)
    x = torch.rand(B, 3, device="cuda", generator=gen) * 2 - 1
    d = torch.randn(B, 3, device="cuda", generator=gen)
    d = d / d.norm(dim=-1, keepdim=True)
    w_s = torch.rand(B, device="cuda", generator=gen)
    w_c = torch.rand(B, 3, device="cuda", generator=gen)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FOC_FUSED_HEAD", mode)
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            sigma, rgb = m(x, d)
        assert sigma.dtype == torch.float32 and sigma.shape == (B,) and rgb.shape == (B, 3)
        ((sigma * w_s).sum() * 1e-2 + (rgb.float() * w_c).sum()).backward()
        out[mode] = (sigma.detach().clone(), rgb.detach().float().clone(), m.sigma_net.weights.grad.clone(), m.color_net.weights.grad.clone(),
                     m.encoder.embeddings.grad.clone())
    a, b = out["1"], out["0"]
    assert torch.allclose(a[0], b[0], rtol=2e-6, atol=0), "sigma = exp(h0) in fp32"
    assert torch.equal(a[1], b[1]), "rgb: half-rounded sigmoid"
    for k, name in ((2, "sigma_net"), (3, "color_net"), (4, "embeddings")):
        scale = b[k].abs().max().item()
        assert (a[k] - b[k]).abs().max().item() <= 4e-3 * scale + 1e-6, name
    # inference mode gives the same values as training mode
    m.eval()
    monkeypatch.setenv("FOC_FUSED_HEAD", "1")
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        s2, r2 = m(x, d)
    assert torch.equal(s2, a[0]) and torch.equal(r2.float(), a[1])


@pytest.mark.parametrize("B", [4096, 1000 + 13])
def test_fused_field_matches_separate_nodes(B, monkeypatch):
    """hashgrid_mlp (encoder output kept as [L,B,C] planes, planar-input MLP kernels) == sigma_net.forward_padded(encoder(x)):
    identical h bits, same parameter gradients (fp32 atomics / fixed-point sums: tolerance)."""
    from focnerf_amd.field import hashgrid_mlp, field_fusable
    m = _model(1, False).train()
    assert field_fusable(m.encoder, m.sigma_net)
    gen = torch.Generator(device="cuda").manual_seed(11)
    x = torch.rand(B, 3, device="cuda", generator=gen) * 2 - 1
    gh = (torch.randn(B, 16, device="cuda", generator=gen) * 0.1).half()
    res = {}
    for mode in ("fused", "separate"):
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            h = hashgrid_mlp(m.encoder, m.sigma_net, x, m.bound) if mode == "fused" else m.sigma_net.forward_padded(m.encoder(x, bound=m.bound))
        h.backward(gh)
        res[mode] = (h.detach().clone(), m.sigma_net.weights.grad.clone(), m.encoder.embeddings.grad.clone())
    assert torch.equal(res["fused"][0], res["separate"][0])
    for k, name in ((1, "mlp weights"), (2, "embeddings")):
        scale = res["separate"][k].abs().max().item()
        assert (res["fused"][k] - res["separate"][k]).abs().max().item() <= 2e-3 * scale + 1e-7, name
    # no-grad / eval calls keep nothing
    m.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        h2 = hashgrid_mlp(m.encoder, m.sigma_net, x, m.bound)
    assert torch.equal(h2, res["fused"][0])


@pytest.mark.parametrize("layers", [(2, 3), (2, 2), (3, 3)])
@pytest.mark.parametrize("M", [5000, 64 * 7 + 1])
def test_fused_inference_kernel_matches_separate_kernels(layers, M, monkeypatch):
    """foc_nerf_field_inference (sigma net -> head -> colour net in one kernel) vs the separate kernels it replaces, per-sample and
    per-ray directions. sigma: same bits. rgb: the geometry features enter the colour net's first MFMA in the chained k order instead
    of the natural one, i.e. the same products summed in another association — at most one fp16 ulp on a handful of values."""
    from focnerf_amd.network import NeRFNetwork
    torch.manual_seed(2)
    m = NeRFNetwork(bound=1, num_layers=layers[0], num_layers_color=layers[1]).cuda().eval()
    m.encoder.embeddings.data.uniform_(-0.5, 0.5)
    gen = torch.Generator(device="cuda").manual_seed(4)
    x = torch.rand(M, 3, device="cuda", generator=gen) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(M, 3, device="cuda", generator=gen), dim=-1)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FOC_FUSED_INFER", mode)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            out[mode] = m(x, d)
    assert torch.equal(out["1"][0], out["0"][0]), "sigma"
    dr = (out["1"][1].float() - out["0"][1].float()).abs()
    assert dr.max().item() <= 4.9e-4 and (dr > 0).float().mean().item() < 1e-3, "rgb"
    # fixed-step renderer in eval mode: whole-field kernel + one-pass tail vs the separate inference kernels
    from focnerf_amd import synthetic
    poses = synthetic.rand_poses(1, "cuda", radius=2.0, generator=torch.Generator().manual_seed(1))
    ro, rd = synthetic.get_rays(poses, synthetic.intrinsics(32, 32), 32, 32)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FOC_FUSED_INFER", mode)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            res[mode] = m.run(ro, rd, None, fused=True, num_steps=96, upsample_steps=0, bg_color=1.0, perturb=False, return_fields=True)
    for k in ("densities", "rgbs", "weights_sum", "depth", "image"):
        a, b = res["1"][k].float(), res["0"][k].float()
        assert a.shape == b.shape, k
        if k == "densities":
            assert torch.equal(a, b), (k, (a - b).abs().max())
        elif k == "rgbs":
            assert (a - b).abs().max().item() <= 4.9e-4 and ((a - b).abs() > 0).float().mean().item() < 1e-3
        else:   # same weights, summed over the ray in a different association
            assert torch.allclose(a, b, atol=2e-5, rtol=0, equal_nan=True), (k, (a - b).abs().max())


def test_sh_encoder_kernel_matches_expression():
    from focnerf_amd.shencoder import SHEncoder, sh_encode_deg4
    enc = SHEncoder()
    d = torch.nn.functional.normalize(torch.randn(7, 333, 3, device="cuda"), dim=-1)
    got = enc(d)
    assert got.shape == (7, 333, 16) and got.dtype == torch.float32
    assert torch.allclose(got, sh_encode_deg4(d), rtol=2e-6, atol=1e-7)
    dg = d.clone().requires_grad_(True)
    enc(dg).sum().backward()                                   # differentiable form still available
    assert dg.grad is not None and torch.isfinite(dg.grad).all()


@pytest.mark.parametrize("path", ["train_forward", "eval_forward", "fused_run_train", "fused_run_eval"])
def test_data_writes_are_seen_by_the_fused_paths(path):
    """Writes through `.data` do not bump a parameter's version counter, and the reference trainer writes that way around every
    evaluation (torch_ema copy_to / restore, nerf/utils.py:1164-1174). The fused paths run on fp16 copies of the table and the weight
    blobs: a copy must never outlive the call that made it (outside an explicit `half_cache_scope`). forward -> p.data.mul_(2) ->
    forward must change; -> p.data.copy_(old) -> forward must give the first result bit for bit."""
    from focnerf_amd import synthetic
    m = _model(1, False)
    rays_o, rays_d = synthetic.make_view_rays(32, 32, 1, 1, seed=3, device="cuda")
    rays_o, rays_d = rays_o[:, :256].contiguous(), rays_d[:, :256].contiguous()
    x = torch.rand(640, 3, device="cuda") * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(640, 3, device="cuda"), dim=-1)

    def evaluate():
        with torch.autocast("cuda", dtype=torch.float16):
            if path == "train_forward":
                m.train()
                s, c = m(x, d)
                return torch.cat([s.float().reshape(-1), c.float().reshape(-1)]).detach().clone()
            if path == "eval_forward":
                m.eval()
                with torch.no_grad():
                    s, c = m(x, d)
                return torch.cat([s.float().reshape(-1), c.float().reshape(-1)]).clone()
            if path == "fused_run_train":
                m.train()
                out = m.run(rays_o, rays_d, num_steps=64, upsample_steps=0, perturb=False, fused=True)
                return out["image"].detach().float().reshape(-1).clone()
            m.eval()
            with torch.no_grad():
                out = m.render(rays_o, rays_d, staged=True, max_ray_batch=128, num_steps=64, upsample_steps=0, perturb=False, fused=True)
            return out["image"].float().reshape(-1).clone()

    for pname in ("encoder.embeddings", "sigma_net.weights", "color_net.weights"):
        p = dict(m.named_parameters())[pname]
        first = evaluate()
        old = p.data.clone()
        v = p._version
        p.data.mul_(2)
        assert p._version == v                                   # the premise: nothing a cache could key on
        changed = evaluate()
        assert not torch.equal(first, changed), f"{path}: a write to {pname}.data was not seen"
        p.data.copy_(old)
        again = evaluate()
        assert torch.equal(first, again), f"{path}: restoring {pname}.data did not restore the result"


def test_half_cache_scope_is_the_only_cache():
    from focnerf_amd.field import _half_of, half_cache_scope
    p = torch.nn.Parameter(torch.rand(1000, 2, device="cuda"))
    a = _half_of(p)
    assert _half_of(p).data_ptr() != a.data_ptr() or True      # fresh conversions (the allocator may recycle the address)
    p.data.mul_(2)
    assert torch.equal(_half_of(p), p.detach().half())
    with half_cache_scope():
        h1 = _half_of(p)
        with half_cache_scope():
            assert _half_of(p) is h1
        assert _half_of(p) is h1
    p.data.mul_(0.5)
    assert torch.equal(_half_of(p), p.detach().half())


def test_native_inference_loop_edge_cases():
    """foc_occ_render_step through run_cuda: jittered first samples (perturb), a handful of rays, rays that all miss the box, and a view
    whose ray count is not a multiple of anything — against the Python loop with the reference's boolean-mask compaction."""
    from focnerf_amd import synthetic
    bound = 2
    m = _model(bound, True, seed=4).eval()
    o, d = synthetic.make_view_rays(40, 40, bound, 1, seed=6, device="cuda")
    kw = dict(staged=False, dt_gamma=1 / 128, max_steps=1024, bg_color=1.0, T_thresh=1e-4)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        for n in (1, 7, 1599):
            a = m.render(o[:, :n], d[:, :n], perturb=False, device_compaction=False, **kw)
            b = m.render(o[:, :n], d[:, :n], perturb=False, device_compaction=True, **kw)
            assert torch.equal(a["image"], b["image"]) and torch.equal(a["depth"], b["depth"]), n
        # rays pointing away from the box: nothing to march, the background comes back
        away = -d[:, :500]
        far_o = o[:, :500] * 4.0
        a = m.render(far_o, away, perturb=False, device_compaction=False, **kw)
        b = m.render(far_o, away, perturb=False, device_compaction=True, **kw)
        assert torch.equal(a["image"], b["image"]) and bool((b["image"] == 1.0).all())
        # jitter: both loop forms draw torch.rand(n) once, first thing — with the same seed they march the SAME jittered first samples, and
        # the jitter belongs to the reference's first iteration only (its composite_rays continues from the un-jittered t): bit for bit
        for seed, mx, th in ((0, 1024, 1e-4), (1, 100, 1e-4), (2, 1024, 0.3)):
            kwp = dict(kw, max_steps=mx, T_thresh=th)
            torch.manual_seed(seed)
            p = m.render(o, d, perturb=True, device_compaction=True, **kwp)
            torch.manual_seed(seed)
            r = m.render(o, d, perturb=True, device_compaction=False, **kwp)
            assert torch.equal(p["image"], r["image"]) and torch.equal(p["depth"], r["depth"]), (seed, mx, th)
        q = m.render(o, d, perturb=False, device_compaction=True, **kw)
        torch.manual_seed(0)
        p = m.render(o, d, perturb=True, device_compaction=True, **kw)
        assert torch.isfinite(p["image"]).all() and not torch.equal(p["image"], q["image"])
        assert (p["image"] - q["image"]).abs().mean() < 2e-2


def test_native_inference_loop_with_the_camera_inside_the_box():
    """Cameras INSIDE the bound (near = min_near, long empty stretches before the object: single advances that more than double t, where the
    re-derived t = last_t + fl(t - last_t) of a wide burst can differ from the march's own t): native loop against the Python loop on the
    reference's schedule, bit for bit, over step caps and thresholds, with cameras close to the object, further out and next to a face of the box."""
    from focnerf_amd import synthetic
    bound = 2
    m = _model(bound, True, seed=7).eval()
    intr = synthetic.intrinsics(56, 56)
    for radius, seed in ((0.9, 3), (1.5, 4), (1.95, 5)):                     # inside the box: close to the object / further out / next to a face
        g = torch.Generator().manual_seed(seed)
        poses = synthetic.rand_poses(1, "cuda", radius=radius, generator=g)
        o, d = synthetic.get_rays(poses, intr, 56, 56)
        assert float(o.abs().max()) < bound                               # the camera is inside the box: every ray starts at min_near
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            for max_steps, thresh in ((1024, 1e-4), (64, 1e-4), (1024, 0.5)):
                kw = dict(staged=False, perturb=False, dt_gamma=1 / 128, max_steps=max_steps, bg_color=1.0, T_thresh=thresh)
                a = m.render(o, d, device_compaction=False, **kw)
                b = m.render(o, d, device_compaction=True, **kw)
                assert torch.equal(a["image"], b["image"]) and torch.equal(a["depth"], b["depth"]), (radius, max_steps, thresh)
        assert (a["image"] < 0.99).any(), radius


@pytest.mark.parametrize("burst", ["1", "3", "8", "16"])
def test_native_inference_loop_burst_lengths_and_the_step_cap(burst, monkeypatch):
    """The native loop deals a ray's samples over iterations in bursts of FOC_RENDER_BURST (default 8) where the reference's rule gives one sample
    per ray while most rays are alive: same image and depth, bit for bit, as the Python loop on the reference's schedule — also when the loop
    ends on `max_steps` (100, 37: not multiples of the burst; every ray is still alive then) and with a transmittance threshold that ends rays
    inside a burst (T_thresh 0.3), an opaque field (density_scale-free: a bias on the density) included."""
    from focnerf_amd import synthetic
    monkeypatch.setenv("FOC_RENDER_BURST", burst)
    bound = 2
    m = _model(bound, True, seed=5).eval()
    o, d = synthetic.make_view_rays(48, 48, bound, 1, seed=8, device="cuda")
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        for max_steps, thresh in ((1024, 1e-4), (100, 1e-4), (37, 1e-4), (1024, 0.3)):
            kw = dict(staged=False, perturb=False, dt_gamma=1 / 128, max_steps=max_steps, bg_color=1.0, T_thresh=thresh)
            a = m.render(o, d, device_compaction=False, **kw)
            b = m.render(o, d, device_compaction=True, **kw)
            assert torch.equal(a["image"], b["image"]) and torch.equal(a["depth"], b["depth"]), (max_steps, thresh)
        assert (a["image"] < 0.99).any()


def test_native_inference_loop_at_baseline_view_size():
    """BASELINE configs[2] render size: one 800 x 800 view (640 000 rays) through the native loop (one C call per iteration, bursts of 8 samples
    per ray, late count) and through the Python loop with the reference's boolean-mask compaction and schedule: the same image and depth, bit for bit."""
    from focnerf_amd import synthetic
    import bench
    bound = 2
    m = _model(bound, True, seed=0).eval()
    poses, intr = bench.make_training_rays(torch.device("cuda"), bound, 1, seed=0)
    o, d = synthetic.get_rays(poses[:1], intr, 800, 800)
    kw = dict(staged=False, perturb=False, dt_gamma=1 / 128, max_steps=1024, bg_color=1.0, T_thresh=1e-4)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        a = m.render(o, d, device_compaction=False, **kw)
        b = m.render(o, d, device_compaction=True, **kw)
    assert a["image"].shape == (1, 640000, 3)
    assert torch.equal(a["image"], b["image"]) and torch.equal(a["depth"], b["depth"])
    assert (a["image"] < 0.99).any()
