"""CPU: the torch restatement timed as BASELINE.json configs[0] (oracle/torch_cpu_nerf.py) against the reference's own code.
cpu_network.npz was produced by the reference's nerf/network.py NeRFNetwork class run through the reference's NeRFRenderer.run
(tests/golden/make_golden.py cpu_network_fixture); the restated network + `run_fixed_steps` must reproduce it — same torch ops in the
same order, so bit for bit on the machine that wrote the fixture, and to the last bits of an sgemm (ISA-dependent blocking) elsewhere."""
import os

import numpy as np
import torch

import oracle
from oracle import torch_cpu_nerf as tcn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load():
    g = np.load(os.path.join(GOLDEN, "cpu_network.npz"))
    nl, base, log2, des = (int(v) for v in g["encoder_cfg"])
    torch.manual_seed(0)
    m = tcn.NeRFNetworkCPU(bound=int(g["bound"]), encoder_kwargs=dict(num_levels=nl, base_resolution=base, log2_hashmap_size=log2))
    # desired_resolution is fixed by the fixture's grid, not by 2048 * bound
    m.encoder = tcn.HashGridCPU(num_levels=nl, base_resolution=base, log2_hashmap_size=log2, desired_resolution=des)
    sd = {k[len("param/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param/")}
    missing, unexpected = m.load_state_dict(sd, strict=True), None
    return g, m


def test_restated_network_and_run_reproduce_the_reference():
    g, m = _load()
    assert np.array_equal(m.encoder.offsets.numpy(), g["param/encoder.offsets"])
    o, d = torch.from_numpy(g["rays_o"]), torch.from_numpy(g["rays_d"])
    T = int(g["T"])
    m.eval()
    with torch.no_grad():
        ev = tcn.run_fixed_steps(m, o[None], d[None], T)
    miss = np.isnan(g["eval_depth"])
    assert miss.sum() > 0 and np.array_equal(np.isnan(ev["depth"][0].numpy()), miss)          # rays that miss the box: 0 * NaN, as in the reference
    np.testing.assert_allclose(ev["densities"].squeeze(-1).numpy(), g["eval_densities"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(ev["image"][0].numpy(), g["eval_image"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(ev["weights_sum"].numpy(), g["eval_weights_sum"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(ev["depth"][0].numpy()[~miss], g["eval_depth"][~miss], rtol=0, atol=2e-6)
    np.testing.assert_allclose(ev["rgbs"].numpy(), g["eval_rgbs"], rtol=0, atol=2e-6)
    # training mode: loss and every parameter gradient
    m.train()
    tr = tcn.run_fixed_steps(m, o[None], d[None], T)
    target = torch.from_numpy(g["train_target"])
    ok = torch.isfinite(tr["depth"][0])
    loss = torch.nn.functional.mse_loss(tr["image"][0][ok], target[ok])
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(g["train_loss"]), rtol=1e-6)
    for i, lin in enumerate(m.sigma_net):
        np.testing.assert_allclose(lin.weight.grad.numpy(), g[f"grad_sigma_net_{i}"], rtol=1e-4, atol=1e-8)
    for i, lin in enumerate(m.color_net):
        np.testing.assert_allclose(lin.weight.grad.numpy(), g[f"grad_color_net_{i}"], rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(m.encoder.embeddings.grad.numpy(), g["grad_embeddings"], rtol=1e-4, atol=1e-9)
    assert np.abs(g["grad_embeddings"]).max() > 0


def test_torch_hash_grid_and_near_far_agree_with_the_c_oracle():
    """HashGridCPU restates gridencoder.cu:87-245 with torch index ops; oracle.c restates it in C. Same indices (dense levels, hashed
    levels, the box faces), same trilinear weights; the values differ only through nvcc's contraction of `x * scale + 0.5` into one FMA,
    which the C oracle spells out and torch's separate multiply and add do not have (<= 1 ulp of a position up to 2048: ~1e-4 of a
    table entry at the finest levels)."""
    torch.manual_seed(0)
    enc = tcn.HashGridCPU(desired_resolution=2048)
    assert np.array_equal(enc.offsets.numpy(), np.load(os.path.join(GOLDEN, "wrappers.npz"))["big_offsets"])   # == the reference GridEncoder's table
    enc.embeddings.data.uniform_(-1, 1)
    x = torch.rand(4000, 3) * 2 - 1
    x[0], x[1], x[2] = 1.0, -1.0, torch.tensor([1.0, -1.0, 0.3])
    with torch.no_grad():
        y = enc(x, bound=1).numpy()
    S = float(np.log2(enc.per_level_scale))
    ref = oracle.grid_encode_forward(((x + 1) / 2).numpy(), enc.embeddings.detach().numpy(), enc.offsets.numpy(), 3, 2, 16, S, 16)
    ref = np.transpose(ref, (1, 0, 2)).reshape(4000, 32)
    assert np.abs(y - ref).max() < 6e-4 and np.abs(y[:, :8] - ref[:, :8]).max() < 2e-5
    out = enc(torch.tensor([[1.5, 0.0, 0.0]]), bound=1)                       # outside the box: zeros (gridencoder.cu:110-135)
    assert torch.all(out == 0)
    o = torch.randn(2000, 3) * 1.5
    d = torch.nn.functional.normalize(torch.randn(2000, 3), dim=-1)
    aabb = torch.tensor([-2.0, -2, -2, 2, 2, 2])
    n, f = tcn.near_far(o, d, aabb, 0.2)
    n2, f2 = oracle.near_far_from_aabb(o.numpy(), d.numpy(), aabb.numpy(), 0.2)
    assert np.array_equal(n.numpy(), n2) and np.array_equal(f.numpy(), f2) and (n2 > 1e30).sum() > 10


def test_baseline_timer_runs_on_a_tiny_sample():
    r = tcn.time_baseline(render_budget_s=0.5, train_steps=1, train_rays=8, side=16, num_steps=16, chunk=64, threads=2)
    assert r["render"]["rays_per_sec"] > 0 and r["train"]["samples_per_sec"] > 0 and r["cores"] == 2 and r["kind"] == "port"
    assert r["extrapolated"] is True and r["train"]["steps_timed"] == 1                    # 8-ray steps are not the protocol's 64-ray (chunk) steps
    full = tcn.time_baseline(train_steps=3, train_rays=64, side=16, num_steps=16, chunk=64, threads=2)
    assert full["extrapolated"] is False and full["train"]["steps_timed"] == 3 and "256 of 256 rays" in full["protocol"]      # the whole view, every step
    short = tcn.time_baseline(train_steps=5, train_rays=64, side=16, num_steps=16, chunk=64, threads=2, budget_s=0.0)          # the safety net: what was reached
    assert short["extrapolated"] is True and short["train"]["steps_timed"] >= 1
