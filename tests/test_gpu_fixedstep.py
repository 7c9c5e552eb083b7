"""GPU parity: the fused fixed-step render path (csrc/fixedstep.hip + focnerf_amd/fixedstep.py) against
(a) the torch restatement of the reference's NeRFRenderer.run on the SAME network (focnerf_amd.renderer.run, itself
    pinned to the reference by tests/golden via the oracle), forward and gradients;
(b) the CPU oracle's fixed-step composite on the fields it produced."""
import numpy as np
import pytest
import torch

import oracle
from util import to_np

pytestmark = pytest.mark.gpu


def _model(bound, seed):
    from focnerf_amd.network import NeRFNetwork
    torch.manual_seed(seed)
    m = NeRFNetwork(bound=bound, cuda_ray=False).cuda()
    m.encoder.embeddings.data.uniform_(-0.5, 0.5)
    return m


def _rays(bound, hw, seed):
    from focnerf_amd import synthetic
    o, d = synthetic.make_view_rays(hw, hw, bound, 1, seed=seed, device="cuda", radius=2.0 * bound)
    return o, d


@pytest.mark.parametrize("bound,T,perturb", [(1, 512, False), (2, 128, True), (1, 65, False)])
def test_fused_matches_torch_run_forward_and_backward(bound, T, perturb):
    from focnerf_amd.fixedstep import render_fixed_steps
    m = _model(bound, 1).train()
    o, d = _rays(bound, 24, 3)
    N = o.shape[1]
    target = torch.rand(1, N, 3, device="cuda")
    outs = {}
    for fused in (False, True):
        m.zero_grad()
        torch.manual_seed(7)                       # same perturbation noise for both paths
        with torch.autocast("cuda", dtype=torch.float16):
            if fused:
                res = render_fixed_steps(m, o, d, num_steps=T, bg_color=1.0, perturb=perturb)
            else:
                res = m.run(o, d, num_steps=T, upsample_steps=0, bg_color=1.0, perturb=perturb)
            loss = ((res["image"] - target) ** 2).mean() + 0.1 * res["depth"].nan_to_num().mean() + 0.05 * res["weights_sum"].mean()
        (loss * 128.0).backward()
        outs[fused] = dict(image=res["image"].detach().clone(), depth=res["depth"].detach().clone(), ws=res["weights_sum"].detach().clone(),
                           g_emb=m.encoder.embeddings.grad.clone(), g_s=m.sigma_net.weights.grad.clone(), g_c=m.color_net.weights.grad.clone())
    a, b = outs[False], outs[True]
    assert torch.allclose(a["image"], b["image"], atol=2e-4), (a["image"] - b["image"]).abs().max()
    assert torch.allclose(a["ws"], b["ws"], atol=2e-5)
    hit = torch.isfinite(a["depth"])
    assert torch.equal(hit, torch.isfinite(b["depth"]))
    assert torch.allclose(a["depth"][hit], b["depth"][hit], atol=2e-5)
    for k, tol in (("g_c", 3e-2), ("g_s", 3e-2), ("g_emb", 3e-2)):
        ga, gb = a[k].float(), b[k].float()
        scale = ga.abs().max().item()
        assert scale > 0 and torch.isfinite(gb).all()
        err = (ga - gb).abs().max().item()
        assert err <= tol * scale, f"{k}: max err {err} vs scale {scale}"
        # direction agreement over the whole tensor
        cos = torch.nn.functional.cosine_similarity(ga.flatten(), gb.flatten(), dim=0).item()
        assert cos > 0.999, f"{k}: cosine {cos}"


def test_fused_fields_match_oracle_composite():
    from focnerf_amd.fixedstep import render_fixed_steps
    from focnerf_amd import raymarching
    m = _model(1, 5).eval()
    o, d = _rays(1, 32, 4)
    # a few rays whose line misses the box: tangent direction at distance 2 > sqrt(3)
    t = torch.cross(o[0, :7], torch.tensor([[0.3, -0.5, 0.8]], device="cuda").expand(7, 3), dim=-1)
    d = d.clone()
    d[0, :7] = t / t.norm(dim=-1, keepdim=True)
    with torch.no_grad():
        res = render_fixed_steps(m, o, d, num_steps=512, bg_color=1.0, perturb=False, return_fields=True)
    nears, fars = raymarching.near_far_from_aabb(o[0], d[0], m.aabb_infer, m.min_near)
    img4, depth = oracle.composite_fixed_steps(to_np(res["densities"].squeeze(-1)), to_np(res["rgbs"]), to_np(nears), to_np(fars), 1.0, clamp01=False)
    hit = to_np(nears) < 1e30
    assert hit.any() and (~hit).any()
    np.testing.assert_allclose(to_np(res["image"][0])[hit], img4[hit, :3], atol=1e-4)
    np.testing.assert_allclose(to_np(res["depth"][0])[hit], depth[hit], atol=1e-4)
    assert np.isnan(to_np(res["depth"][0])[~hit]).all()            # 0/0 depth of rays that miss the box, as in the reference
    np.testing.assert_allclose(to_np(res["image"][0])[~hit], 1.0, atol=1e-6)


@pytest.mark.parametrize("hw,T,perturb,bg", [(32, 512, False, "scalar"), (33, 100, True, "ray"), (9, 65, False, "scalar"), (1, 2, False, "scalar"),
                                             (8, 64, True, "scalar")])
def test_blocked_sample_order_is_bitwise_the_ray_major_path(hw, T, perturb, bg, monkeypatch):
    """The inference path orders its samples in 64-ray blocks between its kernels (fixed_sample -> encoder -> whole-field kernel ->
    composite; include/focnerf.h `ray_block`): a different walk order of the same per-sample arithmetic. Everything the caller sees —
    image, depth, weights_sum, `densities`, `rgbs`, the packed field4 — is bit for bit what the ray-major order (FOC_RAY_BLOCK=0) gives,
    for ray counts that are no multiple of 64 or 16 (padded last block), depth counts that are no multiple of 64, perturbed depths
    (the noise array stays ray-major) and per-ray backgrounds."""
    from focnerf_amd.fixedstep import render_fixed_steps, render_field4, fixed_sample
    from focnerf_amd import raymarching
    m = _model(1, 11).eval()
    o, d = _rays(1, hw, 6)
    N = hw * hw
    bgc = 1.0 if bg == "scalar" else torch.rand(N, 3, device="cuda")
    out = {}
    for rb in ("0", "64"):
        monkeypatch.setenv("FOC_RAY_BLOCK", rb)
        torch.manual_seed(3)
        with torch.no_grad():
            res = render_fixed_steps(m, o, d, num_steps=T, bg_color=bgc, perturb=perturb, return_fields=True)
            f4 = render_field4(m, o[0], d[0], num_steps=T)
        out[rb] = (res, f4)
    a, b = out["0"], out["64"]
    for k in ("image", "weights_sum", "densities", "rgbs"):
        assert torch.equal(a[0][k], b[0][k]), k
    assert torch.equal(torch.nan_to_num(a[0]["depth"]), torch.nan_to_num(b[0]["depth"]))
    assert torch.equal(a[1], b[1])
    assert a[0]["densities"].shape == (N, T, 1) and a[1].shape == (N, T, 4)
    # the sample kernel itself: the blocked rows are the ray-major rows permuted, padding = the last ray
    aabb = m.aabb_infer
    nears, fars = raymarching.near_far_from_aabb(o[0], d[0], aabb, m.min_near)
    noise = torch.rand(N * T, device="cuda") if perturb else None
    rm, _ = fixed_sample(o[0], d[0], nears, fars, aabb, noise, T, m.bound)
    bl, _ = fixed_sample(o[0], d[0], nears, fars, aabb, noise, T, m.bound, ray_block=64)
    nb = -(-N // 64)
    n = torch.arange(nb * 64, device="cuda").clamp(max=N - 1).view(nb, 1, 64)
    rows = (n * T + torch.arange(T, device="cuda").view(1, T, 1)).reshape(-1)
    assert bl.shape == (nb * 64 * T, 3) and torch.equal(torch.nan_to_num(bl), torch.nan_to_num(rm[rows]))


@pytest.mark.parametrize("streams", ["1", "2", "3"])
def test_staged_render_assembles_the_view_in_place_on_alternating_streams(streams, monkeypatch):
    """`render(staged=True)` in eval mode returns image, depth and the per-sample fields of the whole view (nerf/renderer.py:511-560). The
    fused path writes every chunk's part straight into those buffers, chunks alternating over side streams: the result is bit for bit the
    unstaged `run()` on all rays, for a chunk size that divides neither the ray count nor 64, two views in the batch, and a caller that
    is itself on a non-default stream."""
    from focnerf_amd import synthetic
    monkeypatch.setenv("FOC_RENDER_STREAMS", streams)
    m = _model(1, 21).eval()
    o, d = synthetic.make_view_rays(20, 20, 1, 2, seed=8, device="cuda", radius=2.0)      # [2, 400, 3]
    T = 96
    user = torch.cuda.Stream()
    user.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(user), torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        o2, d2 = o * 1.0, d * 1.0                                    # produced on the caller's stream: the side streams must wait for it
        staged = m.render(o2, d2, staged=True, max_ray_batch=150, num_steps=T, upsample_steps=0, perturb=False, fused=True, bg_color=1.0)
        image = staged["image"].clone()                              # consumed on the caller's stream right away
    torch.cuda.current_stream().wait_stream(user)
    assert staged["densities"].shape == (2, 400, T) and staged["rgbs"].shape == (2, 400, T, 3)
    for b in range(2):
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            ref = m.run(o[b:b + 1], d[b:b + 1], num_steps=T, upsample_steps=0, perturb=False, fused=True, bg_color=1.0)
        assert torch.equal(image[b], ref["image"][0]) and torch.equal(staged["image"][b], ref["image"][0])
        assert torch.equal(torch.nan_to_num(staged["depth"][b]), torch.nan_to_num(ref["depth"][0]))
        assert torch.equal(staged["densities"][b], ref["densities"].view(400, T))
        assert torch.equal(staged["rgbs"][b], ref["rgbs"])


def test_sample_positions_bit_exact_vs_torch():
    """fixed_sample reproduces the reference's z_vals / xyz arithmetic bit for bit (incl. torch.linspace's fill order)."""
    from focnerf_amd.fixedstep import fixed_sample
    from focnerf_amd import raymarching
    o, d = _rays(2, 16, 9)
    o, d = o[0], d[0]
    N, T, bound = o.shape[0], 512, 2.0
    aabb = torch.tensor([-bound] * 3 + [bound] * 3, device="cuda")
    nears, fars = raymarching.near_far_from_aabb(o, d, aabb, 0.2)
    noise = torch.rand(N * T, device="cuda")
    for nz in (None, noise):
        enc_in, xyzs = fixed_sample(o, d, nears, fars, aabb, nz, T, bound, want_xyzs=True)
        z = torch.linspace(0.0, 1.0, T, device="cuda").unsqueeze(0).expand(N, T)
        z = nears[:, None] + (fars - nears)[:, None] * z
        if nz is not None:
            z = z + (nz.view(N, T) - 0.5) * ((fars - nears) / T)[:, None]
        ref = o[:, None, :] + d[:, None, :] * z[..., None]
        ref = torch.min(torch.max(ref, aabb[:3]), aabb[3:]).reshape(-1, 3)
        hit = (nears < 1e30).repeat_interleave(T)
        assert torch.equal(xyzs[hit], ref[hit])
        assert torch.equal(enc_in[hit], ((ref + bound) / (2 * bound))[hit])


def test_fixed_step_training_fits_a_scene_fused_and_plain():
    """End to end on FOC's default path: 120 Adam steps (GradScaler, fp16 autocast) on rays of an analytic scene — a red ball on a
    white background — through the fully fused route (encoder->MLP node on [L,B,C] planes, activation re-evaluation, fused head and
    composite, run-merged binned grid backward) and through the torch glue of NeRFRenderer.run with the plain per-op nodes. Both must
    fit the scene; they see the same batches, so their loss curves must also stay close to each other."""
    import os
    from focnerf_amd import synthetic
    from focnerf_amd.network import NeRFNetwork
    bound, T, n_rays = 1, 128, 1024
    gen = torch.Generator().manual_seed(0)
    poses = synthetic.rand_poses(4, "cuda", radius=2.0, generator=gen)
    intr = synthetic.intrinsics(96, 96)
    ro, rd = synthetic.get_rays(poses, intr, 96, 96)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    # analytic target: ray hits the ball of radius 0.45 at the origin -> red, else white
    bq = (ro * rd).sum(-1)
    hit = (bq * bq - ((ro * ro).sum(-1) - 0.45 ** 2)) > 0
    target = torch.ones(ro.shape[0], 3, device="cuda")
    target[hit] = torch.tensor([0.9, 0.1, 0.1], device="cuda")
    batches = [torch.randint(0, ro.shape[0], (n_rays,), generator=gen).cuda() for _ in range(120)]
    curves = {}
    for mode in ("fused", "plain"):
        for k in ("FOC_FUSED_FIELD", "FOC_FUSED_HEAD", "FOC_MLP_RECOMPUTE"):
            os.environ[k] = "1" if mode == "fused" else "0"
        try:
            torch.manual_seed(1)
            m = NeRFNetwork(bound=bound).cuda().train()
            opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
            scaler = torch.amp.GradScaler("cuda")
            losses = []
            for idx in batches:
                with torch.autocast("cuda", dtype=torch.float16):
                    out = m.run(ro[idx][None], rd[idx][None], None, fused=(mode == "fused"), num_steps=T, upsample_steps=0, bg_color=1.0, perturb=True)
                    loss = torch.nn.functional.mse_loss(out["image"], target[idx][None])
                opt.zero_grad()
                scaler.scale(loss).backward()
                scaler.step(opt)
                scaler.update()
                losses.append(loss.item())
            curves[mode] = np.array(losses)
        finally:
            for k in ("FOC_FUSED_FIELD", "FOC_FUSED_HEAD", "FOC_MLP_RECOMPUTE"):
                os.environ.pop(k, None)
    for mode, c in curves.items():
        assert np.isfinite(c).all(), mode
        assert c[-10:].mean() < 0.2 * c[:5].mean(), f"{mode}: loss {c[:5].mean():.4f} -> {c[-10:].mean():.4f}"
    assert abs(curves["fused"][-10:].mean() - curves["plain"][-10:].mean()) < 0.5 * curves["plain"][-10:].mean() + 1e-3


def test_whole_step_replays_as_a_hip_graph():
    """focnerf_amd.graph.GraphedStep: forward, backward, GradScaler and fused Adam of the fixed-step path captured once and replayed;
    the kernels of the C ABI run on torch's capturing stream and the scratch buffers are persistent, so the replayed steps follow
    the eager ones."""
    from focnerf_amd import synthetic
    from focnerf_amd.graph import GraphedStep
    from focnerf_amd.network import NeRFNetwork
    gen = torch.Generator().manual_seed(0)
    poses = synthetic.rand_poses(2, "cuda", radius=2.0, generator=gen)
    ro, rd = synthetic.get_rays(poses, synthetic.intrinsics(48, 48), 48, 48)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    target = torch.rand(ro.shape[0], 3, generator=gen).cuda()
    batches = [torch.randint(0, ro.shape[0], (512,), generator=gen).cuda() for _ in range(6)]

    def make():
        torch.manual_seed(3)
        m = NeRFNetwork(bound=1).cuda().train()
        opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True, capturable=True)
        sc = torch.amp.GradScaler("cuda", init_scale=1024.0)

        def step(o, d, t):
            with torch.autocast("cuda", dtype=torch.float16):
                out = m.run(o[None], d[None], None, fused=True, num_steps=64, upsample_steps=0, bg_color=1.0, perturb=False)
                loss = torch.nn.functional.mse_loss(out["image"], t[None])
            opt.zero_grad(set_to_none=False)
            sc.scale(loss).backward()
            sc.step(opt)
            sc.update()
            return loss.detach()
        return m, step

    m_e, step_e = make()
    m_g, step_g = make()
    eager = [float(step_e(ro[i], rd[i], target[i])) for i in batches]
    # first step eagerly on the second model too (creates the scratch buffers and the cached host copy of the level offsets: neither
    # may happen inside a capture); the capture itself executes nothing, the remaining five batches are replays
    got = [float(step_g(ro[batches[0]], rd[batches[0]], target[batches[0]]))]
    graphed = GraphedStep(step_g, (ro[batches[1]], rd[batches[1]], target[batches[1]]), warmup=0)
    got += [float(graphed(ro[i], rd[i], target[i])) for i in batches[1:]]
    assert np.allclose(got, eager, rtol=2e-2, atol=1e-4), (got, eager)
    assert torch.allclose(m_g.sigma_net.weights, m_e.sigma_net.weights, atol=2e-3)
    assert not torch.equal(m_g.sigma_net.weights, make()[0].sigma_net.weights), "the replayed steps did update the parameters"


def test_amp_overflow_steps_are_detected_and_skipped():
    """A GradScaler that starts far too high overflows the fp16 gradients of the first steps. The inf/NaN must reach the parameter
    gradients (the binned grid backward has no encoding for them in its fixed-point sums and poisons the rows instead), so that the
    scaler skips those steps and backs off; the parameters stay finite and training proceeds once the scale is sane."""
    from focnerf_amd import synthetic
    from focnerf_amd.network import NeRFNetwork
    torch.manual_seed(0)
    m = NeRFNetwork(bound=1).cuda().train()
    opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 40, backoff_factor=1.0 / 65536.0, growth_interval=10 ** 6)
    gen = torch.Generator().manual_seed(0)
    poses = synthetic.rand_poses(1, "cuda", radius=2.0, generator=gen)
    ro, rd = synthetic.get_rays(poses, synthetic.intrinsics(32, 32), 32, 32)
    target = torch.rand(1, ro.shape[1], 3, generator=gen).cuda()
    before = [p.detach().clone() for p in m.parameters()]
    scales, losses = [], []
    for it in range(6):
        with torch.autocast("cuda", dtype=torch.float16):
            out = m.run(ro, rd, None, fused=True, num_steps=64, upsample_steps=0, bg_color=1.0, perturb=False)
            loss = torch.nn.functional.mse_loss(out["image"], target)
        opt.zero_grad(set_to_none=True)
        scaler.scale(loss).backward()
        if it == 0:
            assert not all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None), "the overflow must be visible in the gradients"
            assert not torch.isfinite(m.encoder.embeddings.grad).all(), "in the hash-table gradient too"
        scaler.step(opt)
        scaler.update()
        scales.append(scaler.get_scale())
        losses.append(loss.item())
        if it == 0:
            assert all(torch.equal(a, b) for a, b in zip(before, m.parameters())), "an overflowed step must not touch the parameters"
    assert scales[0] < 2.0 ** 40 and scales[-1] <= scales[0]
    assert all(torch.isfinite(p).all() for p in m.parameters())
    assert any(not torch.equal(a, b) for a, b in zip(before, m.parameters())), "later steps did train"
    assert np.isfinite(losses).all()


@pytest.mark.parametrize("c_width", [4, 16])
@pytest.mark.parametrize("N,T,layers,perturb,bg", [(96, 512, 3, False, "scalar"), (37, 65, 3, True, "ray"), (50, 128, 2, False, "scalar"), (1, 2, 3, False, "scalar"),
                                                   (41, 64, 3, False, "wide")])
def test_render_tail_node_is_bitwise_the_three_node_chain(N, T, layers, perturb, bg, c_width, monkeypatch):
    """`_render_tail` (density head -> colour network fed from h and a per-ray SH row -> composite, one autograd node) against
    `_density_head` -> FFMLP.forward_padded -> `_fixed_composite`, which materialise the colour network's [M,32] input and its gradient:
    image, weights_sum, depth, sigma, weights, colour logits and grad_h must be the SAME BITS (every value of the colour input sits at
    the k position it has in the materialised row); the weight gradient differs only by the order of its fp32 atomics."""
    from focnerf_amd.ffmlp import FFMLP
    from focnerf_amd import fixedstep
    from focnerf_amd.fixedstep import _density_head, _fixed_composite, _render_tail, ray_sh_rows
    monkeypatch.setattr(fixedstep, "_C_WIDTH", c_width)        # colour logits as [M,4] (default) or full [M,16] rows: both ABI forms
    g = torch.Generator(device="cuda").manual_seed(N * 1000 + T)
    M = N * T
    h0 = (torch.randn(M, 16, generator=g, device="cuda") * 0.7).half()
    h0[:, 0] = (torch.randn(M, generator=g, device="cuda") * 2.0 - 1.0).half()
    if bg == "wide":
        # density logits beyond +-15, where trunc_exp's backward clamps (activation.py:15): the tail takes that factor as
        # clamp(sigma, exp(-15), exp(15)) instead of exp(clamp(h0, -15, 15)) — the same bits only because expf is monotonic; the
        # three-node chain computes the literal form (k_fs_head_bwd)
        h0[::5, 0] = (torch.rand(h0[::5, 0].shape, generator=g, device="cuda") * 12.0 - 24.0).half()        # [-24, -12]
        h0[3::11, 0] = (torch.rand(h0[3::11, 0].shape, generator=g, device="cuda") * 3.0 + 13.5).half()     # [13.5, 16.5]
        h0[0, 0], h0[1, 0], h0[2, 0] = 15.0, -15.0, 15.0078125
    rays_d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g, device="cuda"), dim=-1)
    nears = torch.rand(N, generator=g, device="cuda") * 0.5 + 0.2
    fars = nears + 1.0 + torch.rand(N, generator=g, device="cuda")
    noise = torch.rand(M, generator=g, device="cuda") if perturb else None
    bg_ray = torch.rand(N, 3, generator=g, device="cuda") if bg == "ray" else None
    bg_scalar = 0.0 if bg == "ray" else 1.0
    net = FFMLP(32, 3, 64, layers).cuda().train()
    net.weights.data = torch.randn(net.weights.shape, generator=g, device="cuda") * 0.2
    g_img = torch.randn(N, 3, generator=g, device="cuda")
    g_ws = torch.randn(N, generator=g, device="cuda") * 0.1
    g_dp = torch.randn(N, generator=g, device="cuda") * 0.1
    out = {}
    for fused in (False, True):
        h = h0.clone().requires_grad_(True)
        net.weights.grad = None
        with torch.autocast("cuda", dtype=torch.float16):
            if fused:
                image, ws, depth, sigma, weights, c = _render_tail.apply(h, net.weights, ray_sh_rows(rays_d), nears, fars, noise, bg_ray, bg_scalar, N, T, 1.0, 1e-4,
                                                                         net.num_layers, net.activation)
            else:
                weights, ws, depth, sigma, cin = _density_head.apply(h, rays_d, nears, fars, noise, N, T, 1.0, None)
                c = net.forward_padded(cin)
                image = _fixed_composite.apply(c, weights, bg_ray, bg_scalar, N, T, 1e-4)
        torch.autograd.backward([image, ws, depth], [g_img, g_ws, g_dp])
        out[fused] = dict(image=image.detach(), ws=ws.detach(), depth=depth.detach(), sigma=sigma.detach(), weights=weights.detach(), c=c.detach(),
                          g_h=h.grad.clone(), g_w=net.weights.grad.clone())
    a, b = out[False], out[True]
    a["c"] = a["c"][:, :b["c"].shape[1]]          # the fused tail keeps only the columns that are read: rgb logits + one pad
    for k in ("image", "ws", "depth", "sigma", "weights", "c", "g_h"):
        assert torch.equal(torch.nan_to_num(a[k].float(), nan=12345.0), torch.nan_to_num(b[k].float(), nan=12345.0)), \
            f"{k}: {(a[k].float() - b[k].float()).abs().max().item()}"
        assert torch.equal(torch.isnan(a[k]), torch.isnan(b[k])), k
    assert a["g_h"].abs().max() > 0 and a["g_h"][:, 1:].abs().max() > 0
    scale = a["g_w"].abs().max().item()
    assert scale > 0 and (a["g_w"] - b["g_w"]).abs().max().item() <= 2e-3 * scale


def test_render_tail_is_the_path_render_fixed_steps_trains_through(monkeypatch):
    """render_fixed_steps with and without the fused tail (FOC_FUSED_TAIL=0): same image bits; gradients equal up to atomics order."""
    from focnerf_amd.fixedstep import render_fixed_steps, tail_fusable
    m = _model(1, 5).train()
    assert tail_fusable(m)
    o, d = _rays(1, 16, 2)
    target = torch.rand(1, o.shape[1], 3, device="cuda")
    outs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("FOC_FUSED_TAIL", flag)
        m.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16):
            res = render_fixed_steps(m, o, d, num_steps=128, bg_color=1.0, perturb=False)
            loss = ((res["image"] - target) ** 2).mean()
        (loss * 1024.0).backward()
        outs[flag] = (res["image"].detach().clone(), res["depth"].detach().clone(), m.encoder.embeddings.grad.clone(), m.sigma_net.weights.grad.clone(),
                      m.color_net.weights.grad.clone())
    a, b = outs["0"], outs["1"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1].nan_to_num(), b[1].nan_to_num())
    for ga, gb in zip(a[2:], b[2:]):
        scale = ga.abs().max().item()
        assert scale > 0 and (ga - gb).abs().max().item() <= 5e-3 * scale


def test_outside_mask_criterion_trains_through_the_fused_tail():
    """nerf/renderer.py:163-165: in training with YOLO details the renderer also returns the norm of the densities outside the object
    mask. On the fused path h feeds both that criterion (torch) and the fused tail node; the two gradients must add up as in run()."""
    from focnerf_amd.fixedstep import render_fixed_steps
    m = _model(1, 4).train()
    o, d = _rays(1, 12, 6)
    N, T = o.shape[1], 64
    mask = (torch.rand(1, N, T, device="cuda") > 0.5)
    yolo = (mask, None, None)
    target = torch.rand(1, N, 3, device="cuda")
    outs = {}
    for fused in (False, True):
        m.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16):
            res = render_fixed_steps(m, o, d, yolo, num_steps=T, bg_color=1.0, perturb=False) if fused else \
                m.run(o, d, yolo, num_steps=T, upsample_steps=0, bg_color=1.0, perturb=False, fused=False)
            crit = res["criterion_outside_mask"]
            loss = ((res["image"] - target) ** 2).mean() + 1e-3 * crit
        (loss * 256.0).backward()
        outs[fused] = (res["image"].detach().clone(), crit.detach().clone(), m.encoder.embeddings.grad.clone(), m.sigma_net.weights.grad.clone(),
                       m.color_net.weights.grad.clone())
    a, b = outs[False], outs[True]
    assert torch.allclose(a[0], b[0], atol=2e-4)
    assert torch.allclose(a[1].float(), b[1].float(), rtol=1e-3)
    for ga, gb in zip(a[2:], b[2:]):
        scale = ga.abs().max().item()
        assert scale > 0 and (ga.float() - gb.float()).abs().max().item() <= 3e-2 * scale
