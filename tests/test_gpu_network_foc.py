"""GPU: the FOC object-conditioned network (focnerf_amd/network_foc.py; reference nerf/network_tcnn.py:451-681, whose arithmetic lives in the
un-vendored tinycudann — parity unpinned by construction) on the fused fast paths.

The object feature is one 16-vector per object: inside the kernels its share of the colour network's first layer is a per-neuron
constant (csrc/ffmlp.hip, MlpHead). Checked here against
  (a) the CPU oracle's chain on the MATERIALISED 48-wide colour input [SH16 | h[1:16] | obj 16 | 0]:
      oracle.grid_encode_forward -> oracle.ffmlp_forward -> oracle/torch_cpu_nerf.sh_encode_deg4 -> 48-wide oracle.ffmlp_forward / _backward
      (fp32 accumulation, fp16 layer outputs), in half-ulps;
  (b) the general path of this library, which materialises that input (`_density_head` / `sample_head` with cin_width 48 -> FFMLP);
  (c) network.NeRFNetwork when the object feature is zero.
"""

import os

import numpy as np
import pytest
import torch

import oracle
from oracle import torch_cpu_nerf
from util import assert_half_close, half_ulp, to_np

pytestmark = pytest.mark.gpu


def _foc(seed=0):
    from focnerf_amd.network_foc import NeRFNetwork
    torch.manual_seed(seed)
    m = NeRFNetwork(bound=1).cuda()
    m.encoder.embeddings.data.uniform_(-0.5, 0.5)
    return m


def _rays(n, seed):
    from focnerf_amd import synthetic
    gen = torch.Generator().manual_seed(seed)
    poses = synthetic.rand_poses(2, "cuda", radius=2.0, generator=gen)
    intr = synthetic.intrinsics(64, 64)
    ro, rd = synthetic.get_rays(poses[:1], intr, 64, 64)
    idx = torch.randperm(64 * 64, generator=gen)[:n].cuda()
    return ro[:, idx].contiguous(), rd[:, idx].contiguous()


def _sh16(dirs):
    """[N,3] torch (any device) -> [N,16] fp16 numpy: the ORACLE's degree-4 SH (oracle/torch_cpu_nerf.py, pinned by the reference-generated
    cpu_network.npz), rounded to half like the colour network's input."""
    return torch_cpu_nerf.sh_encode_deg4(dirs.detach().cpu().float()).numpy().astype(np.float16)


def _cin48(sh_rows, h, obj16):
    """Materialised colour input [SH16 | h[:,1:16] | obj | 0] (fp16 numpy)."""
    B = h.shape[0]
    return np.concatenate([sh_rows, h[:, 1:], np.broadcast_to(obj16[None], (B, 16)), np.zeros((B, 1), np.float16)], 1).astype(np.float16)


def test_foc_network_shapes_and_params():
    m = _foc()
    assert m.color_net.input_dim == 48 and m.sigma_net.input_dim == 32 and m.color_in == 47
    names = {n for n, _ in m.named_parameters()}
    assert {"encoder.embeddings", "sigma_net.weights", "color_net.weights", "yolo_feat_encoder.l0.weight", "yolo_feat_encoder.l1.weight"} <= names
    assert len(m.get_params(1e-2)) == 5
    from focnerf_amd.field import infer_fusable
    from focnerf_amd.fixedstep import tail_fusable
    assert tail_fusable(m) and infer_fusable(m), "the object-conditioned network must be on the fused paths"


@pytest.mark.parametrize("layers,out_width,T,N", [(2, 4, 64, 37), (3, 16, 128, 16), (2, 16, 1, 700), (2, 4, 512, 9)])
def test_color_head_with_object_feature_vs_oracle(layers, out_width, T, N):
    """foc_color_head_forward / _backward with obj_feat through the C ABI against the oracle's 48-wide FFMLP on the materialised input:
    logits and grad_h in half-ulps, every block of grad_weights (the object columns 31..46 = colsum(delta_0) (x) obj, the pad column 0),
    and grad_obj = W0[:,31:47]^T colsum(delta_0)."""
    from focnerf_amd._lib import lib, ptr, stream_of, check
    from focnerf_amd.fixedstep import ray_sh_rows
    g = torch.Generator(device="cuda").manual_seed(layers * 100 + T)
    B = N * T
    h = (torch.randn(B, 16, generator=g, device="cuda") * 0.7).half()
    dirs = torch.nn.functional.normalize(torch.randn(N, 3, generator=g, device="cuda"), dim=-1)
    ray_sh = ray_sh_rows(dirs)
    obj = (torch.randn(16, generator=g, device="cuda") * 0.8).half()
    n_w = 64 * (48 + 64 * (layers - 1) + 16)
    W = (torch.randn(n_w, generator=g, device="cuda") * 0.2).half()
    grad = torch.zeros(B, out_width, dtype=torch.float16, device="cuda")
    grad[:, :3] = (torch.randn(B, 3, generator=g, device="cuda") * 0.05).half()
    g_h0 = (torch.randn(B, generator=g, device="cuda") * 0.05).half()

    out = torch.empty(B, out_width, dtype=torch.float16, device="cuda")
    st = stream_of(h)
    check(lib.foc_color_head_forward(ptr(h), ptr(ray_sh), T, ptr(W), B, 64, layers, 0, ptr(out), out_width, ptr(obj), st), "fwd")
    grad_h = torch.empty(B, 16, dtype=torch.float16, device="cuda")
    g_w = torch.empty(n_w, dtype=torch.float16, device="cuda")
    g_obj = torch.empty(16, dtype=torch.float32, device="cuda")
    ws = torch.empty(lib.foc_ffmlp_backward_workspace_bytes(48, 64, layers), dtype=torch.uint8, device="cuda")
    check(lib.foc_color_head_backward(ptr(grad), ptr(h), ptr(ray_sh), T, ptr(g_h0), ptr(W), B, 64, layers, 0, ptr(grad_h), ptr(g_w), ptr(ws), ws.numel(), out_width,
                                      ptr(obj), ptr(g_obj), st), "bwd")
    torch.cuda.synchronize()

    # ---- oracle on the materialised rows (the SH rows the kernel read: the per-ray table, itself checked against the oracle's SH below)
    sh_np = to_np(ray_sh)
    np.testing.assert_allclose(sh_np.astype(np.float32), _sh16(dirs).astype(np.float32), atol=2e-3)
    cin = _cin48(np.repeat(sh_np, T, axis=0), to_np(h), to_np(obj))
    Wn = to_np(W)
    c_ref, fb = oracle.ffmlp_forward(cin, Wn, 48, 64, layers, 0)
    # logits: fp32 sums in another order -> the half results agree up to rare one-ulp flips that the later layers pass on
    assert_half_close(to_np(out)[:, :3], c_ref[:, :3], ulps=2.0, atol=2e-3, what="colour logits")
    exact = (to_np(out)[:, :3] == c_ref[:, :3]).mean()
    assert exact > 0.97, f"only {exact:.3f} of the logits are the oracle's bits"
    g16 = np.zeros((B, 16), np.float16)
    g16[:, :out_width] = to_np(grad)
    gw_ref, gi_ref, bb = oracle.ffmlp_backward(g16, cin, Wn, fb, 48, 64, layers, 0, True)
    # grad_h = [grad_h0 | grad_cin[:, 16:31]]
    gh = to_np(grad_h).astype(np.float32)
    np.testing.assert_array_equal(gh[:, 0], to_np(g_h0).astype(np.float32))
    scale = np.abs(gi_ref[:, 16:31].astype(np.float32)).max()
    assert scale > 0
    assert np.abs(gh[:, 1:] - gi_ref[:, 16:31].astype(np.float32)).max() <= 4e-3 * scale
    # weight gradients, block by block (fp32 sums over B rows rounded once; the oracle sums in row order)
    gw, gwr = to_np(g_w).astype(np.float32), gw_ref.astype(np.float32)
    w0, w0r = gw[:64 * 48].reshape(64, 48), gwr[:64 * 48].reshape(64, 48)
    for name, a, b in (("dW0[:, :31] (SH + geometry)", w0[:, :31], w0r[:, :31]), ("dW0[:, 31:47] (object feature)", w0[:, 31:47], w0r[:, 31:47]),
                       ("hidden + output matrices", gw[64 * 48:], gwr[64 * 48:])):
        s = np.abs(b).max()
        assert s > 0 and np.abs(a - b).max() <= 3e-3 * s, f"{name}: {np.abs(a - b).max()} vs scale {s}"
    assert not w0[:, 47].any(), "the pad column's input is zero: so is its weight gradient"
    # grad_obj against float64 from the oracle's own delta_0 (backward_buffer[last] = gradient of forward layer 0)
    d0 = bb[layers - 1].astype(np.float64)
    ref_obj = d0.sum(0) @ Wn[:64 * 48].reshape(64, 48)[:, 31:47].astype(np.float64)
    s = np.abs(ref_obj).max()
    assert s > 0 and np.abs(to_np(g_obj) - ref_obj).max() <= 3e-3 * s


@pytest.mark.parametrize("N,T,perturb,bg", [(96, 128, False, "scalar"), (37, 65, True, "ray"), (1, 2, False, "scalar")])
def test_render_tail_with_object_feature_vs_materialised_chain(N, T, perturb, bg):
    """`_render_tail(..., obj_feat)` against `_density_head(cin 48)` -> FFMLP(48).forward_padded -> `_fixed_composite`: density-side outputs are
    the same bits (the object feature does not touch them), colour logits agree in half-ulps (the feature's products are summed first), image
    1e-4, and the gradients of h, the weights and the object feature agree."""
    from focnerf_amd.ffmlp import FFMLP
    from focnerf_amd.fixedstep import _density_head, _fixed_composite, _render_tail, ray_sh_rows
    g = torch.Generator(device="cuda").manual_seed(N * 1000 + T)
    M = N * T
    h0 = (torch.randn(M, 16, generator=g, device="cuda") * 0.7).half()
    h0[:, 0] = (torch.randn(M, generator=g, device="cuda") * 2.0 - 1.0).half()
    rays_d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g, device="cuda"), dim=-1)
    nears = torch.rand(N, generator=g, device="cuda") * 0.5 + 0.2
    fars = nears + 1.0 + torch.rand(N, generator=g, device="cuda")
    noise = torch.rand(M, generator=g, device="cuda") if perturb else None
    bg_ray = torch.rand(N, 3, generator=g, device="cuda") if bg == "ray" else None
    bg_scalar = 0.0 if bg == "ray" else 1.0
    net = FFMLP(48, 3, 64, 2).cuda().train()
    net.weights.data = torch.randn(net.weights.shape, generator=g, device="cuda") * 0.2
    obj0 = torch.randn(16, generator=g, device="cuda") * 0.8
    g_img = torch.randn(N, 3, generator=g, device="cuda")
    g_ws = torch.randn(N, generator=g, device="cuda") * 0.1
    g_dp = torch.randn(N, generator=g, device="cuda") * 0.1
    out = {}
    for fused in (False, True):
        h = h0.clone().requires_grad_(True)
        obj = obj0.clone().requires_grad_(True)
        net.weights.grad = None
        with torch.autocast("cuda", dtype=torch.float16):
            if fused:
                image, ws, depth, sigma, weights, c = _render_tail.apply(h, net.weights, ray_sh_rows(rays_d), nears, fars, noise, bg_ray, bg_scalar, N, T, 1.0, 1e-4,
                                                                         net.num_layers, net.activation, obj)
            else:
                weights, ws, depth, sigma, cin = _density_head.apply(h, rays_d, nears, fars, noise, N, T, 1.0, obj)
                assert cin.shape == (M, 48)
                c = net.forward_padded(cin)
                image = _fixed_composite.apply(c, weights, bg_ray, bg_scalar, N, T, 1e-4)
        torch.autograd.backward([image, ws, depth], [g_img, g_ws, g_dp])
        out[fused] = dict(image=image.detach(), ws=ws.detach(), depth=depth.detach(), sigma=sigma.detach(), weights=weights.detach(), c=c.detach()[:, :3],
                          g_h=h.grad.clone(), g_w=net.weights.grad.clone(), g_obj=obj.grad.clone())
    a, b = out[False], out[True]
    for k in ("ws", "depth", "sigma", "weights"):
        assert torch.equal(torch.nan_to_num(a[k].float(), nan=12345.0), torch.nan_to_num(b[k].float(), nan=12345.0)), k
    assert_half_close(to_np(b["c"]), to_np(a["c"]), ulps=2.0, atol=2e-3, what="colour logits")
    assert torch.allclose(a["image"], b["image"], atol=1e-4), (a["image"] - b["image"]).abs().max()
    for k, tol in (("g_h", 4e-3), ("g_w", 3e-3)):
        s = a[k].float().abs().max().item()
        assert s > 0 and (a[k].float() - b[k].float()).abs().max().item() <= tol * s, k
    # the general path sums half-rounded per-sample gradients of the object columns; the fused path sums delta_0 in fp32 first
    s = a["g_obj"].abs().max().item()
    assert s > 0 and (a["g_obj"] - b["g_obj"]).abs().max().item() <= 1e-2 * s + 1e-6 * M
    gw = b["g_w"][:64 * 48].view(64, 48)
    assert gw[:, 31:47].abs().max() > 0 and not gw[:, 47].any()


@pytest.mark.parametrize("blocked", [False, True])
def test_foc_field_inference_vs_oracle_chain(blocked):
    """The whole-field kernel with an object feature (k_nerf_infer<..., OBJ>): sigma and rgb against the oracle chain
    grid_encode_forward -> ffmlp_forward(32->64->64->16) -> [oracle SH | geo | obj | 0] -> ffmlp_forward(48->64->64->16) -> sigmoid,
    per-sample directions and the staged render's 64-ray block order."""
    from focnerf_amd.field import field_infer
    m = _foc(3).eval()
    m.color_net.weights.data.mul_(1.5)
    obj = (torch.randn(16, device="cuda") * 0.8)
    T = 8
    N = 200 if blocked else 1000
    B = (-(-N // 64) * 64 * T) if blocked else N
    xn = torch.rand(B, 3, device="cuda")
    d = torch.nn.functional.normalize(torch.randn(N, 3, device="cuda"), dim=-1)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        sigma, rgb = field_infer(m, xn, d, dir_div=T if blocked else 1, dir_block=64 if blocked else 0, obj_feat=obj)
    S = float(np.log2(m.encoder.per_level_scale))
    enc = oracle.grid_encode_forward(to_np(xn), to_np(m.encoder.embeddings).astype(np.float16), to_np(m.encoder.offsets), 3, 2, 16, S, 16)
    enc = np.ascontiguousarray(np.transpose(enc, (1, 0, 2)).reshape(B, 32))
    h = oracle.ffmlp_forward(enc, to_np(m.sigma_net.weights).astype(np.float16), 32, 64, 2, 0, training=False)
    if blocked:
        r = np.arange(B)
        ray = np.minimum((r // (64 * T)) * 64 + r % 64, N - 1)
    else:
        ray = np.arange(B)
    cin = _cin48(_sh16(d)[ray], h, to_np(obj.half()))
    c = oracle.ffmlp_forward(cin, to_np(m.color_net.weights).astype(np.float16), 48, 64, 2, 0, training=False)[:, :3]
    # log-density in half-ulps of the logit (sigma = exp(h0): one half-ulp of h0 at |h0| ~ 4 is 0.4 % of sigma)
    h0_gpu = np.log(to_np(sigma))
    # the logit is a sum of 64 products of ~0.1-sized terms: a one-ulp flip of one hidden activation moves it by ~1e-5 whatever its own size
    assert_half_close(h0_gpu, h[:, 0], ulps=2.0, atol=1e-4, what="density logit")
    assert (np.abs(h0_gpu - h[:, 0].astype(np.float32)) <= 3e-6 * np.maximum(1, np.abs(h[:, 0].astype(np.float32)))).mean() > 0.97, "most logits are the oracle's bits"
    rgb_ref = (1.0 / (1.0 + np.exp(-c.astype(np.float32)))).astype(np.float16).astype(np.float32)
    assert_half_close(to_np(rgb), rgb_ref, ulps=2.0, atol=1e-6, what="rgb")
    assert (to_np(rgb) == rgb_ref).mean() > 0.97


def test_foc_fixed_step_fused_matches_torch():
    """run(fused=True) — the fused tail with the object feature — against run(fused=False) (torch glue of NeRFRenderer.run): image, depth,
    outside-mask criterion, and the gradients of every parameter group including the YOLO feature encoder."""
    m = _foc().train()
    N, T = 96, 128
    ro, rd = _rays(N, 3)
    gen = torch.Generator().manual_seed(5)
    mask = (torch.rand(1, N, generator=gen) > 0.4).cuda()
    feat = torch.randn(144, generator=gen).numpy().astype(np.float32)
    yolo = (mask, None, feat)
    target = torch.rand(1, N, 3, generator=gen).cuda()
    out = {}
    for fused in (True, False):
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            res = m.run(ro, rd, yolo, fused=fused, num_steps=T, upsample_steps=0, bg_color=1.0, perturb=False)
            loss = ((res['image'] - target) ** 2).mean() + 1e-3 * res['criterion_outside_mask']
        loss.backward()
        out[fused] = (res['image'].detach().clone(), res['depth'].detach().clone(), res['criterion_outside_mask'].detach().clone(),
                      {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
        assert 'timing' in res and 'densities' not in res
    a, b = out[True], out[False]
    assert torch.allclose(a[0], b[0], atol=5e-4), (a[0] - b[0]).abs().max()
    assert torch.allclose(a[1], b[1], atol=5e-4, equal_nan=True)
    assert torch.allclose(a[2], b[2], rtol=1e-3)
    assert set(a[3]) == set(b[3])
    for n in a[3]:
        scale = b[3][n].abs().max().item()
        assert (a[3][n] - b[3][n]).abs().max().item() <= 3e-2 * scale + 1e-6, n
        assert scale > 0 or "encoder_dir" in n, f"{n} received no gradient"


def test_foc_fused_tail_equals_the_materialised_path_through_render_fixed_steps(monkeypatch):
    """render_fixed_steps on the FOC network with the fused tail and with FOC_FUSED_TAIL=0 (cin [M,48] materialised)."""
    from focnerf_amd.fixedstep import render_fixed_steps
    m = _foc(2).train()
    ro, rd = _rays(64, 4)
    feat = np.random.default_rng(1).standard_normal(144).astype(np.float32)
    target = torch.rand(1, 64, 3, device="cuda")
    outs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("FOC_FUSED_TAIL", flag)
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            res = render_fixed_steps(m, ro, rd, (None, None, feat), num_steps=128, bg_color=1.0, perturb=False)
            loss = ((res["image"] - target) ** 2).mean()
        (loss * 1024.0).backward()
        outs[flag] = (res["image"].detach().clone(), res["depth"].detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    a, b = outs["0"], outs["1"]
    assert torch.allclose(a[0], b[0], atol=1e-4) and torch.equal(a[1].nan_to_num(), b[1].nan_to_num())
    assert set(a[2]) == set(b[2]) and "yolo_feat_encoder.l0.weight" in a[2]
    for n in a[2]:
        scale = a[2][n].abs().max().item()
        assert scale > 0 and (a[2][n] - b[2][n]).abs().max().item() <= 1e-2 * scale, n


def test_foc_eval_returns_fields_and_matches_plain_network_with_zero_object(monkeypatch):
    from focnerf_amd.network import NeRFNetwork as PlainNetwork
    m = _foc().eval()
    torch.manual_seed(0)
    p = PlainNetwork(bound=1, num_layers_color=2).cuda().eval()
    # same field: copy the shared parameters; colour net: the plain 32-wide input is the 48-wide one without the object columns
    p.encoder.embeddings.data.copy_(m.encoder.embeddings.data)
    p.sigma_net.weights.data.copy_(m.sigma_net.weights.data)
    H = 64
    w48 = m.color_net.weights.data
    first48 = w48[:H * 48].view(H, 48)
    p.color_net.weights.data.copy_(torch.cat([torch.cat([first48[:, :31], first48[:, 47:48]], dim=1).reshape(-1), w48[H * 48:]]))
    N, T = 64, 64
    ro, rd = _rays(N, 9)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        a = m.run(ro, rd, None, fused=True, num_steps=T, upsample_steps=0, bg_color=1.0, perturb=False)
        b = p.run(ro, rd, None, fused=True, num_steps=T, upsample_steps=0, bg_color=1.0, perturb=False)
    assert a['densities'].shape == (N, T, 1) and a['rgbs'].shape == (N, T, 3)       # eval mode: the reference's result dictionary
    assert torch.equal(a['densities'], b['densities'])
    # a zero object feature adds an exact zero to every layer-0 accumulator: the same bits as the 32-wide network
    assert torch.equal(a['rgbs'], b['rgbs']) and torch.equal(a['image'], b['image'])
    # forward(x, d, yolo_details) on sample lists: whole-field kernel (48-wide) vs the materialised path vs torch expressions
    x = torch.rand(3000, 3, device="cuda") * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(3000, 3, device="cuda"), dim=-1)
    obj = torch.randn(16, device="cuda")
    outs = {}
    for name, env in (("infer", {}), ("head", {"FOC_FUSED_INFER": "0"}), ("torch", {"FOC_FUSED_HEAD": "0"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            outs[name] = m(x, d, (None, None, obj))
        for k in env:
            monkeypatch.delenv(k)
    assert torch.equal(outs["infer"][0], outs["head"][0])
    assert torch.allclose(outs["head"][0], outs["torch"][0], rtol=2e-6)
    assert_half_close(to_np(outs["infer"][1].float()), to_np(outs["head"][1].float()), ulps=2.0, what="rgb: whole-field kernel vs materialised input")
    assert torch.allclose(outs["head"][1].float(), outs["torch"][1].float(), atol=1e-3)


def test_foc_objects_through_render_field4_and_the_combiner(monkeypatch):
    """COMBINED.py:592-618 with the network it actually constructs (network_tcnn topology, :84): K FOC objects, each with its own object
    feature, through `render_field4` (fused: whole-field kernel with the feature) and the select + composite — against the same objects with
    FOC_FUSED_INFER=0 (materialised 48-wide input, general kernels)."""
    from focnerf_amd import raymarching as rm
    from focnerf_amd.combine import ObjectCombiner
    from focnerf_amd.field import half_cache_scope
    from focnerf_amd.fixedstep import render_field4
    K, T = 3, 64
    objs = [_foc(10 + k).eval() for k in range(K)]
    rng = np.random.default_rng(3)
    feats = [rng.standard_normal(144).astype(np.float32) for _ in range(K)]
    ro, rd = _rays(300, 12)
    vo, vd = ro[0].contiguous(), rd[0].contiguous()
    vn, vf = rm.near_far_from_aabb(vo, vd, objs[0].aabb_infer, objs[0].min_near)
    imgs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("FOC_FUSED_INFER", flag)
        fns = [(lambda lo, hi, out, m=m, f=f: render_field4(m, vo[lo:hi], vd[lo:hi], num_steps=T, yolo_details=(None, None, f), out=out)) for m, f in zip(objs, feats)]
        comb = ObjectCombiner(rank=0, world_size=1)
        with torch.no_grad(), half_cache_scope():
            f4 = render_field4(objs[0], vo, vd, num_steps=T, yolo_details=(None, None, feats[0])).clone()
            img, dep = comb.render_view(fns, vo.shape[0], vn, vf, T, max_ray_batch=128)
        imgs[flag] = (f4, img.clone(), dep.clone())
    a, b = imgs["1"], imgs["0"]
    assert torch.equal(a[0][..., 0], b[0][..., 0]), "densities do not depend on the object feature path"
    assert_half_close(to_np(a[0][..., 1:]), to_np(b[0][..., 1:]), ulps=2.0, what="packed rgb")
    assert torch.allclose(a[1], b[1], atol=1e-4) and torch.allclose(a[2], b[2], atol=1e-4, equal_nan=True)
    # the object feature matters: another feature, another colour
    with torch.no_grad(), half_cache_scope():
        other = render_field4(objs[0], vo, vd, num_steps=T, yolo_details=(None, None, feats[1]))
    assert (other[..., 1:] - a[0][..., 1:]).abs().max() > 1e-3


def test_object_feature_encoder_vector_path_equals_the_linear_layers():
    """`_TinyMLP` evaluates the one-vector case as matrix-vector products in fp32 (network_foc._tiny_mlp_vec): same values and gradients as
    its two bias-free linear layers, with and without autocast."""
    from focnerf_amd.network_foc import _TinyMLP
    torch.manual_seed(4)
    m = _TinyMLP(144, 16).cuda()
    x0 = torch.randn(1, 144, device="cuda")
    gy = torch.randn(1, 16, device="cuda")
    res = {}
    for name in ("vector", "linear", "vector_autocast"):
        x = x0.clone().requires_grad_(True)
        m.zero_grad(set_to_none=True)
        if name == "linear":
            y = m.l1(torch.relu(m.l0(x)))
        elif name == "vector":
            y = m(x)
        else:
            with torch.autocast("cuda", dtype=torch.float16):
                y = m(x)
        assert y.dtype == torch.float32 and y.shape == (1, 16)
        y.backward(gy)
        res[name] = (y.detach().clone(), x.grad.clone(), m.l0.weight.grad.clone(), m.l1.weight.grad.clone())
    for name in ("vector", "vector_autocast"):
        for a, b in zip(res[name], res["linear"]):
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-5), name
    # a batch of vectors takes the linear layers
    assert m(torch.randn(3, 144, device="cuda")).shape == (3, 16)


def test_foc_fused_step_at_baseline_size_rows_are_those_of_the_materialised_path(monkeypatch):
    """BASELINE configs[1] size — 4096 rays x 512 samples — on the FOC network: the fused tail's image rows for a random subset of the rays
    must be what the materialised path (cin [M,48] in memory, general kernels) gives for those rays alone (rays are independent of each
    other: a size-independent property), and the gradient of the object-feature encoder must be the batch-linear sum of two half batches."""
    from focnerf_amd.fixedstep import render_fixed_steps
    m = _foc(7).train()
    from focnerf_amd import synthetic
    gen = torch.Generator().manual_seed(2)
    poses = synthetic.rand_poses(1, "cuda", radius=2.0, generator=gen)
    ro, rd = synthetic.get_rays(poses, synthetic.intrinsics(800, 800), 800, 800, torch.randint(0, 640000, (1, 4096), generator=gen).cuda())
    feat = torch.randn(144, generator=gen).cuda()
    T = 512
    with torch.autocast("cuda", dtype=torch.float16):
        full = render_fixed_steps(m, ro, rd, (None, None, feat), num_steps=T, bg_color=1.0, perturb=False)
    pick = torch.randperm(4096, generator=gen)[:48].cuda()
    monkeypatch.setenv("FOC_FUSED_TAIL", "0")
    with torch.autocast("cuda", dtype=torch.float16):
        part = render_fixed_steps(m, ro[:, pick], rd[:, pick], (None, None, feat), num_steps=T, bg_color=1.0, perturb=False)
    monkeypatch.delenv("FOC_FUSED_TAIL")
    assert torch.allclose(full["image"][0, pick], part["image"][0], atol=1e-4), (full["image"][0, pick] - part["image"][0]).abs().max()
    assert torch.equal(full["depth"][0, pick].nan_to_num(), part["depth"][0].nan_to_num())
    # batch linearity of the parameter gradients that the colsum trick produces (object columns of dW0, the encoder's weights)
    target = torch.rand(1, 4096, 3, device="cuda")

    def grads(sel):
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            res = render_fixed_steps(m, ro[:, sel], rd[:, sel], (None, None, feat), num_steps=T, bg_color=1.0, perturb=False)
            loss = ((res["image"] - target[:, sel]) ** 2).sum()
        (loss * 64.0).backward()
        w0 = m.color_net.weights.grad[:64 * 48].view(64, 48)[:, 31:47].float().clone()
        return w0, m.yolo_feat_encoder.l1.weight.grad.clone()
    a, b, c = grads(slice(0, 2048)), grads(slice(2048, 4096)), grads(slice(0, 4096))
    for x, y, z, name in ((a[0], b[0], c[0], "dW0[:, 31:47]"), (a[1], b[1], c[1], "object-feature encoder")):
        s = z.abs().max().item()
        assert s > 0 and (x + y - z).abs().max().item() <= 1e-2 * s, name


def test_occ_render_step_with_object_feature():
    """foc_occ_render_step's obj_feat argument (FOC's object-conditioned colour network inside the native occupancy-render iteration): the
    iteration's sigma / rgb equal the whole-field call with the object feature on the samples the step marched, and the composite is that of
    composite_rays on them — one iteration on a half-full occupancy grid, ray-major and sample-major sample arrays."""
    from focnerf_amd import raymarching as rm, synthetic
    from focnerf_amd._lib import lib, ptr, stream_of, check
    from focnerf_amd.field import field_infer, _half_of, half_cache_scope
    m = _foc(9).eval()
    m.color_net.weights.data.mul_(1.5)
    obj = (torch.randn(16, device="cuda") * 0.8).half()
    o, d = synthetic.make_view_rays(40, 40, 1, 1, seed=4, device="cuda", radius=2.5)
    o, d = o.view(-1, 3).contiguous(), d.view(-1, 3).contiguous()
    n, burst, C, H = o.shape[0], 8, 1, 128
    gen = torch.Generator().manual_seed(0)
    grid = (torch.rand(C, H ** 3, generator=gen) < 0.5).float().cuda()
    bits = rm.packbits(grid, 0.5)
    aabb = torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32, device="cuda")
    near, far = rm.near_far_from_aabb(o, d, aabb, 0.2)
    enc, sn, cn = m.encoder, m.sigma_net, m.color_net
    L = enc.offsets.shape[0] - 1
    M = n * burst
    S = float(np.log2(enc.per_level_scale))
    results = {}
    from focnerf_amd._lib import option
    for sm in ("1", "0"):
        with option("FOC_OCC_SAMPLE_MAJOR", int(sm)):
            samples = torch.full((M * 8,), float("nan"), device="cuda")
            planes = torch.empty(L * M * 2, dtype=torch.float16, device="cuda")
            sigma, rgb = torch.empty(M, device="cuda"), torch.empty(M * 3, device="cuda")
            alive, out = torch.arange(n, dtype=torch.int32, device="cuda"), torch.empty(n, dtype=torch.int32, device="cuda")
            count = torch.empty(1, dtype=torch.int32, device="cuda")
            scratch = torch.empty(lib.foc_occ_render_step_scratch_bytes(n), dtype=torch.uint8, device="cuda")
            t_now = near.clone()
            ws, dp, im = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(n, 3, device="cuda")
            still = torch.zeros(n, device="cuda")
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16), half_cache_scope():
                emb, wsn, wcn = _half_of(enc.embeddings), _half_of(sn.weights), _half_of(cn.weights)
                check(lib.foc_occ_render_step(n, burst, ptr(alive), ptr(out), ptr(count), ptr(t_now), ptr(o), ptr(d), 1.0, 1 / 128, 1024, C, H, ptr(bits), ptr(near),
                                              ptr(far), ptr(still), ptr(samples), ptr(planes), ptr(sigma), ptr(rgb), ptr(emb), ptr(enc.offsets), None, L, S,
                                              enc.base_resolution, ptr(wsn), sn.num_layers, ptr(wcn), cn.num_layers, sn.activation, ptr(obj), 1e-4, ptr(ws), ptr(dp),
                                              ptr(im), ptr(scratch), 0, None, 0, 1, stream_of(o)), "occ_render_step")
                xn, dirs, dl = samples[: 3 * M].view(M, 3), samples[3 * M: 6 * M].view(M, 3), samples[6 * M:].view(M, 2)
                assert torch.isfinite(samples).all(), "every slot of every entry is written"
                real = dl[:, 0] > 0
                assert 0.2 < real.float().mean() < 1.0
                want_s, want_c = field_infer(m, xn.contiguous(), dirs.contiguous(), obj_feat=obj)
            assert torch.equal(sigma[real], want_s[real]) and torch.equal(rgb.view(M, 3)[real], want_c[real])
            # composite_rays on the step's own samples (ray-major copies of them)
            if sm == "1":
                to_rm = lambda a, k: a.view(burst, n, k).transpose(0, 1).reshape(M, k).contiguous()
            else:
                to_rm = lambda a, k: a.view(M, k)
            alive2, t2 = torch.arange(n, dtype=torch.int32, device="cuda"), near.clone()
            ws2, dp2, im2 = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(n, 3, device="cuda")
            rm.composite_rays(n, burst, alive2, t2, to_rm(sigma, 1).view(M), to_rm(rgb, 3), to_rm(dl, 2), ws2, dp2, im2, 1e-4)
            for a, b in ((ws, ws2), (dp, dp2), (im, im2), (t_now, t2)):
                assert torch.equal(a, b)
            assert int(count) == int((alive2 >= 0).sum()) and ws.max() > 0.05
            results[sm] = (ws, dp, im)
    for a, b in zip(results["1"], results["0"]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B", [1, 31, 64, 65, 129, 8191])
def test_field_forward_train_small_and_ragged_batches(B):
    """The same bitwise comparison on batches below / at / just above one 64-row tile and one workgroup's share (the prefetch of the tile behind
    the last one is clamped to the batch: nothing past row B - 1 may be read into a result or written)."""
    from focnerf_amd._lib import lib, ptr, stream_of, check
    g = torch.Generator(device="cuda").manual_seed(77 + B)
    T = 5
    n_rays = (B + T - 1) // T
    planes = ((torch.rand(16, B, 2, generator=g, device="cuda") - 0.5) * 2).half()
    w_s = (torch.randn(64 * (32 + 64 + 16), generator=g, device="cuda") * 0.25).half()
    w_c = (torch.randn(64 * (32 + 128 + 16), generator=g, device="cuda") * 0.25).half()
    ray_sh = (torch.randn(n_rays, 16, generator=g, device="cuda") * 0.5).half()
    st = stream_of(planes)
    h1 = torch.empty(B, 16, dtype=torch.float16, device="cuda")
    c1 = torch.empty(B, 4, dtype=torch.float16, device="cuda")
    check(lib.foc_ffmlp_forward_planar(ptr(planes), ptr(w_s), B, 32, 16, 64, 2, 0, 6, ptr(h1), st), "sigma forward")
    check(lib.foc_color_head_forward(ptr(h1), ptr(ray_sh), T, ptr(w_c), B, 64, 3, 0, ptr(c1), 4, None, st), "colour forward")
    h2 = torch.full((B + 64, 16), 5.0, dtype=torch.float16, device="cuda")
    c2 = torch.full((B + 64, 4), 5.0, dtype=torch.float16, device="cuda")
    check(lib.foc_field_forward_train(ptr(planes), ptr(w_s), 2, ptr(ray_sh), T, ptr(w_c), 3, 64, 0, B, ptr(h2), ptr(c2), 4, None, st), "fused forward")
    torch.cuda.synchronize()
    assert torch.all(h2[B:] == 5.0) and torch.all(c2[B:] == 5.0), "rows past B were written"
    assert torch.equal(h2[:B].view(torch.int16), h1.view(torch.int16)) and torch.equal(c2[:B].view(torch.int16), c1.view(torch.int16))


@pytest.mark.parametrize("nls,nlc", [(2, 2), (2, 3), (3, 3)])
@pytest.mark.parametrize("out_width,T,with_obj,act", [(4, 512, False, 0), (16, 1, False, 0), (4, 7, True, 0), (16, 64, False, 6)])
def test_field_forward_train_is_bitwise_the_two_calls(nls, nlc, out_width, T, with_obj, act):
    """foc_field_forward_train (round 5, csrc/field_fwd.hip: sigma network on the planes -> h -> colour network fed from h and the per-ray SH rows, one
    kernel) against foc_ffmlp_forward_planar + foc_color_head_forward: h AND the colour logits bit for bit — ragged batch (the last tile and the last
    ray are partial), one SH row per sample / per 7 / per 512 samples, [B,4] and [B,16] logits, FOC's object feature, activation none. The colour
    network's second k-chunk (columns 1..16 of the h row) is assembled from the accumulator tile with v_permlane32_swap instead of re-read from
    memory: this test is what pins that exchange."""
    from focnerf_amd._lib import lib, ptr, stream_of, check
    g = torch.Generator(device="cuda").manual_seed(1000 * nls + 100 * nlc + T)
    B = 40000 + 17
    n_rays = (B + T - 1) // T
    planes = ((torch.rand(16, B, 2, generator=g, device="cuda") - 0.5) * 2).half()
    w_s = (torch.randn(64 * (32 + 64 * (nls - 1) + 16), generator=g, device="cuda") * 0.25).half()
    ld0 = 48 if with_obj else 32
    w_c = (torch.randn(64 * (ld0 + 64 * (nlc - 1) + 16), generator=g, device="cuda") * 0.25).half()
    ray_sh = (torch.randn(n_rays, 16, generator=g, device="cuda") * 0.5).half()
    obj = (torch.randn(16, generator=g, device="cuda") * 0.8).half() if with_obj else None
    st = stream_of(planes)
    h1 = torch.empty(B, 16, dtype=torch.float16, device="cuda")
    c1 = torch.full((B, out_width), 3.0, dtype=torch.float16, device="cuda")
    check(lib.foc_ffmlp_forward_planar(ptr(planes), ptr(w_s), B, 32, 16, 64, nls, act, 6, ptr(h1), st), "sigma forward")
    check(lib.foc_color_head_forward(ptr(h1), ptr(ray_sh), T, ptr(w_c), B, 64, nlc, act, ptr(c1), out_width, ptr(obj), st), "colour forward")
    h2 = torch.full((B + 8, 16), 5.0, dtype=torch.float16, device="cuda")
    c2 = torch.full((B + 8, out_width), 5.0, dtype=torch.float16, device="cuda")
    check(lib.foc_field_forward_train(ptr(planes), ptr(w_s), nls, ptr(ray_sh), T, ptr(w_c), nlc, 64, act, B, ptr(h2), ptr(c2), out_width, ptr(obj), st), "fused forward")
    torch.cuda.synchronize()
    assert torch.all(h2[B:] == 5.0) and torch.all(c2[B:] == 5.0), "rows past B were written"
    assert h1.float().abs().max() > 0.5 and c1.float().abs().max() > 0.1, "degenerate test data"
    assert torch.equal(h2[:B].view(torch.int16), h1.view(torch.int16)), "h differs from foc_ffmlp_forward_planar"
    assert torch.equal(c2[:B].view(torch.int16), c1.view(torch.int16)), "colour logits differ from foc_color_head_forward"
    # refusals on the host
    assert lib.foc_field_forward_train(ptr(planes), ptr(w_s), 4, ptr(ray_sh), T, ptr(w_c), 3, 64, 0, B, ptr(h2), ptr(c2), out_width, None, st) == 1
    assert lib.foc_field_forward_train(ptr(planes), ptr(w_s), 2, ptr(ray_sh), T, ptr(w_c), 3, 32, 0, B, ptr(h2), ptr(c2), out_width, None, st) == 1
