"""GPU: the FOC object-conditioned network (focnerf_amd/network_foc.py, reference nerf/network_tcnn.py:451-681 without
tinycudann — parity unpinned by construction): its fused kernels against its own torch expressions, and against
network.NeRFNetwork when the object feature is zero."""
import numpy as np
import pytest
import torch

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


def test_foc_network_shapes_and_params():
    m = _foc()
    assert m.color_net.input_dim == 48 and m.sigma_net.input_dim == 32 and m.color_in == 47
    names = {n for n, _ in m.named_parameters()}
    assert {"encoder.embeddings", "sigma_net.weights", "color_net.weights", "yolo_feat_encoder.l0.weight", "yolo_feat_encoder.l1.weight"} <= names
    assert len(m.get_params(1e-2)) == 5


def test_foc_fixed_step_fused_matches_torch():
    """run(fused=True) == run(fused=False): image, depth, outside-mask criterion, and the gradients of every parameter group
    including the YOLO feature encoder (whose gradient is a column sum over all samples)."""
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
    assert torch.allclose(a[0], b[0], atol=2e-3), (a[0] - b[0]).abs().max()
    assert torch.allclose(a[1], b[1], atol=2e-3, equal_nan=True)
    assert torch.allclose(a[2], b[2], rtol=1e-3)
    assert set(a[3]) == set(b[3])
    for n in a[3]:
        scale = b[3][n].abs().max().item()
        assert (a[3][n] - b[3][n]).abs().max().item() <= 3e-2 * scale + 1e-6, n
        assert scale > 0 or "encoder_dir" in n, f"{n} received no gradient"


def test_foc_eval_returns_fields_and_matches_plain_network_with_zero_object():
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
    assert torch.allclose(a['image'], b['image'], atol=1e-3)
    # forward(x, d, yolo_details) on sample lists: fused head (48-wide) vs torch expressions
    x = torch.rand(3000, 3, device="cuda") * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(3000, 3, device="cuda"), dim=-1)
    obj = torch.randn(16, device="cuda")
    import os
    outs = {}
    for mode in ("1", "0"):
        os.environ["FOC_FUSED_HEAD"] = mode
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            outs[mode] = m(x, d, (None, None, obj))
    os.environ.pop("FOC_FUSED_HEAD")
    assert torch.allclose(outs["1"][0], outs["0"][0], rtol=2e-6)
    assert torch.allclose(outs["1"][1].float(), outs["0"][1].float(), atol=1e-3)
