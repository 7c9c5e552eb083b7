"""GPU: degenerate sizes through the Python operator API — empty batches, single elements, rays that all miss, a masked colour
query with an empty mask, error behaviour on bad arguments (RuntimeError with the library's message, like TORCH_CHECK)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(bound=1, cuda_ray=False):
    from focnerf_amd.network import NeRFNetwork
    torch.manual_seed(0)
    return NeRFNetwork(bound=bound, cuda_ray=cuda_ray).cuda()


def test_empty_batches_everywhere():
    from focnerf_amd import raymarching
    from focnerf_amd.gridencoder import GridEncoder
    from focnerf_amd.freqencoder import FreqEncoder
    from focnerf_amd.ffmlp import FFMLP
    dev = "cuda"
    z3 = torch.zeros(0, 3, device=dev)
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1], device=dev)
    n, f = raymarching.near_far_from_aabb(z3, z3, aabb, 0.2)
    assert n.shape == (0,) and f.shape == (0,)
    assert raymarching.morton3D(torch.zeros(0, 3, dtype=torch.int32, device=dev)).shape == (0,)
    assert raymarching.morton3D_invert(torch.zeros(0, dtype=torch.int32, device=dev)).shape == (0, 3)
    enc = GridEncoder(desired_resolution=2048).cuda()
    with torch.autocast("cuda", dtype=torch.float16):
        e = enc(z3, bound=1)
        assert e.shape == (0, 32)
        mlp = FFMLP(32, 16, 64, 2).cuda().train()
        xin = torch.zeros(0, 32, device=dev, requires_grad=True)
        y = mlp(xin)
        assert y.shape == (0, 16)
        y.sum().backward()                       # empty backward: zero weight gradient, no launch on zero rows
        assert torch.all(mlp.weights.grad == 0)
    fe = FreqEncoder(input_dim=3, degree=4).cuda()
    assert fe(z3).shape == (0, 27)
    m = _net().eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        s, c = m(z3, z3)
        assert s.shape == (0,) and c.shape == (0, 3)
        out = m.run(z3[None], z3[None], None, fused=True, num_steps=16, upsample_steps=0)
        assert out["image"].shape == (1, 0, 3)


def test_single_sample_and_single_ray():
    m = _net().eval()
    x = torch.tensor([[0.1, -0.2, 0.3]], device="cuda")
    d = torch.tensor([[0.0, 0.0, 1.0]], device="cuda")
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        s1, c1 = m(x, d)
        s2, c2 = m(x.repeat(65, 1), d.repeat(65, 1))
    assert s1.shape == (1,) and torch.isfinite(s1).all() and torch.equal(s2, s1.expand(65)) and torch.equal(c2, c1.expand(65, 3))
    ro = torch.tensor([[[0.0, 0.0, -2.0]]], device="cuda")
    rd = torch.tensor([[[0.0, 0.0, 1.0]]], device="cuda")
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        a = m.run(ro, rd, None, fused=True, num_steps=8, upsample_steps=0, bg_color=1.0)
        b = m.run(ro, rd, None, fused=False, num_steps=8, upsample_steps=0, bg_color=1.0)
    assert torch.allclose(a["image"], b["image"], atol=2e-3) and torch.allclose(a["depth"], b["depth"], atol=2e-3)


def test_rays_that_miss_the_box():
    """near = far = FLT_MAX from near_far_from_aabb (raymarching.cu:139-142): the fixed-step renderer then produces the background
    and a NaN depth exactly like the torch code does ((z - near) / (far - near) = 0/0), the occupancy path no samples."""
    m = _net(2, cuda_ray=True).train()
    from focnerf_amd import synthetic
    m.set_density_grid(synthetic.analytic_density_grid(2, device="cuda"))
    ro = torch.tensor([[[10.0, 10.0, 10.0], [8.0, 9.0, 10.0]]], device="cuda")
    rd = torch.nn.functional.normalize(torch.tensor([[[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]], device="cuda"), dim=-1)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        out = m.render(ro, rd, None, staged=False, perturb=False, force_all_rays=True, dt_gamma=1 / 128, max_steps=64, bg_color=1.0)
    assert torch.allclose(out["image"], torch.ones_like(out["image"]))
    assert int(m.step_counter[(m.local_step - 1) % 16, 0].item()) == 0
    f = _net(1).eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        a = f.run(ro, rd, None, fused=True, num_steps=8, upsample_steps=0, bg_color=1.0)
        b = f.run(ro, rd, None, fused=False, num_steps=8, upsample_steps=0, bg_color=1.0)
    assert torch.equal(torch.isnan(a["depth"]), torch.isnan(b["depth"]))


def test_colour_query_with_empty_mask_and_full_mask():
    m = _net().eval()
    x = torch.rand(10, 3, device="cuda") * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(10, 3, device="cuda"), dim=-1)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        dens = m.density(x)
        none = m.color(x, d, mask=torch.zeros(10, dtype=torch.bool, device="cuda"), **dens)
        full = m.color(x, d, mask=torch.ones(10, dtype=torch.bool, device="cuda"), **dens)
        plain = m.color(x, d, **dens)
    assert torch.all(none == 0) and torch.allclose(full.float(), plain.float())


def test_bad_arguments_raise_runtime_error():
    from focnerf_amd.backend import _gridencoder, _ffmlp
    from focnerf_amd.ffmlp import FFMLP
    x = torch.rand(4, 3, device="cuda")
    emb = torch.rand(64, 2, device="cuda")
    off = torch.tensor([0, 64], dtype=torch.int32, device="cuda")
    out = torch.empty(1, 4, 2, device="cuda")
    with pytest.raises(RuntimeError, match="D must be 2, 3, 4 or 5"):
        _gridencoder.grid_encode_forward(torch.rand(4, 6, device="cuda"), emb, off, out, 4, 6, 2, 1, 0.0, 8, None, 0, False, 0)
    with pytest.raises(RuntimeError, match="C must be 1, 2, 4, or 8"):
        _gridencoder.grid_encode_forward(x, torch.rand(64, 3, device="cuda"), off, torch.empty(1, 4, 3, device="cuda"), 4, 3, 3, 1, 0.0, 8, None, 0, False, 0)
    with pytest.raises(RuntimeError):
        _gridencoder.grid_encode_forward(x.cpu(), emb, off, out, 4, 3, 2, 1, 0.0, 8, None, 0, False, 0)        # CPU tensor: no CPU implementation
    with pytest.raises(AssertionError):
        FFMLP(30, 3, 64, 2)                                                                                     # ffmlp.py:84
    w = torch.zeros(64 * (32 + 64 + 16), dtype=torch.float16, device="cuda")
    with pytest.raises(RuntimeError, match="hidden_dim"):
        _ffmlp.ffmlp_inference(torch.zeros(4, 32, dtype=torch.float16, device="cuda"), w, 4, 32, 16, 512, 2, 0, 6, None,
                               torch.empty(4, 16, dtype=torch.float16, device="cuda"))


def test_graph_keeps_its_scratch_when_a_later_call_needs_more():
    """A captured step bakes in the raw addresses of the library's scratch buffers (march strip, binned-backward workspace, MLP dW
    workspace). A later eager call that needs a larger buffer must not hand the captured one back to the allocator: capture a training
    step, run the same ops eagerly on a much larger batch (and churn the allocator), replay, compare with the eager result."""
    from focnerf_amd import backend
    from focnerf_amd.graph import GraphedStep
    from focnerf_amd.gridencoder import GridEncoder
    from focnerf_amd.ffmlp import FFMLP
    from focnerf_amd.field import hashgrid_mlp
    torch.manual_seed(0)
    enc = GridEncoder(desired_resolution=2048).cuda()
    enc.embeddings.data.uniform_(-0.5, 0.5)
    mlp = FFMLP(32, 16, 64, 2).cuda().train()
    B = 4096
    x = torch.rand(B, 3, device="cuda") * 2 - 1
    gy = (torch.randn(B, 16, device="cuda") * 0.01).half()

    def step(xx, gg):
        enc.embeddings.grad = None
        mlp.weights.grad = None
        with torch.autocast("cuda", dtype=torch.float16):
            h = hashgrid_mlp(enc, mlp, xx, 1)
        h.backward(gg)
        return enc.embeddings.grad, mlp.weights.grad

    ge_ref, gw_ref = (t.clone() for t in step(x, gy))
    g = GraphedStep(step, (x, gy))
    pinned_before = len(backend._scratch._pinned)
    captured = [e[0].data_ptr() for e in backend._scratch._bufs.values() if e[1]]
    assert captured, "the capture saw no scratch buffer: the test no longer exercises what it claims"
    # same stream as the capture (GraphedStep's side stream is private, so go through the scratch of every stream): a 16x larger batch
    for key in list(backend._scratch._bufs):
        ent = backend._scratch._bufs[key]
        if ent[1]:
            s = torch.cuda.ExternalStream(key[2])
            with torch.cuda.stream(s):
                backend._scratch.get(key[0], ent[0].numel() * 16, torch.device("cuda", key[1]))
    torch.cuda.synchronize()
    assert len(backend._scratch._pinned) > pinned_before      # superseded buffers a graph saw are kept, not freed
    big = torch.rand(16 * B, 3, device="cuda") * 2 - 1
    step(big, (torch.randn(16 * B, 16, device="cuda") * 0.01).half())
    junk = [torch.full((1 << 22,), 7.0, device="cuda") for _ in range(64)]      # would land in any block the scratch had given back
    torch.cuda.synchronize()
    del junk
    ge, gw = g(x, gy)
    torch.cuda.synchronize()
    torch.testing.assert_close(gw.float(), gw_ref.float(), rtol=2e-2, atol=2e-3)      # dW: fp32 atomics across workgroups, order dependent
    assert torch.equal(ge, ge_ref)                              # the binned backward is deterministic (fixed-point sums)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_ops_follow_their_tensors_device():
    """An op on cuda:1 tensors while cuda:0 is current (K resident objects spread over the GPUs of one process): the Python binding runs
    every entry point with the device of ITS TENSORS current (`_lib._on_tensor_device`; torch's default stream is the null handle on every
    device, so the library's own FocDeviceGuard cannot learn the device from the stream) and restores the caller's."""
    from focnerf_amd.network import NeRFNetwork
    torch.manual_seed(0)
    m0 = NeRFNetwork(bound=1).to("cuda:0").eval()
    m0.encoder.embeddings.data.uniform_(-0.5, 0.5)
    m1 = NeRFNetwork(bound=1).to("cuda:1").eval()
    m1.load_state_dict(m0.state_dict())
    x = torch.rand(1000, 3) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(1000, 3), dim=-1)
    torch.cuda.set_device(0)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        s0, c0 = m0(x.to("cuda:0"), d.to("cuda:0"))
        s1, c1 = m1(x.to("cuda:1"), d.to("cuda:1"))
    assert torch.cuda.current_device() == 0
    assert s1.device.index == 1 and torch.equal(s0.cpu(), s1.cpu()) and torch.equal(c0.cpu(), c1.cpu())


def test_a_c_caller_with_hip_runtime_buffers_matches_the_oracle(tmp_path):
    """tests/c_abi_consumer.cpp, mode "gpu": hipMalloc'ed buffers, raw pointers and sizes, the NULL stream — near_far_from_aabb (hits, misses,
    origins inside the box, axis-parallel rays), morton3D and packbits against the C oracle linked into the same program, bit for bit; an error
    in between leaves a message and the next call works. No torch and no Python inside that process."""
    import subprocess
    from util import build_c_abi_consumer
    exe = build_c_abi_consumer(tmp_path)
    r = subprocess.run([exe, "gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "C_ABI_CONSUMER_OK gpu" in r.stdout, r.stdout + r.stderr


def test_bench_line_has_every_leg_and_no_leg_reports_an_error():
    """`python bench.py` as the driver runs it (fewer steps, no CPU baseline): ONE JSON line whose legs all measured something — a leg that raises is
    caught inside bench.py and reported as {"error": ...}, which a green exit code would hide (round 5: a local variable of the roofline block shadowed
    the scene's `bound` and took `combined_render`, the leg the N-rank scaling run is about, down with a TypeError)."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--steps", "6", "--warmup", "3", "--no-cpu-baseline", "--render-views", "2", "--combined-views", "1"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])

    def errors(node, path=""):
        found = []
        if isinstance(node, dict):
            for k, v in node.items():
                if k in ("error", "kernel_only_error"):
                    found.append(f"{path}/{k}: {str(v)[:300]}")
                else:
                    found += errors(v, f"{path}/{k}")
        return found
    assert not errors(r), "\n".join(errors(r))
    for leg in ("roofline", "kernels", "render", "render_with_fields", "dropin_ops_path", "graph_replay", "occupancy_path", "render_occupancy", "combined_render", "foc_network"):
        assert leg in r, f"{leg} missing from the bench line: {sorted(r)}"
    assert r["value"] > 1e8 and r["roofline"]["frac"] > 0.05 and len(r["kernels"]) >= 8
    assert "mfma" in r["roofline"] and r["roofline"]["mfma"]["mlp_backward"]["frac"] > 0.05
    assert r["roofline"]["render"]["hbm_real_frac"] is None or r["roofline"]["render"]["hbm_real_frac"] < r["roofline"]["render"]["frac"]
    assert r["dropin_ops_path"]["render"]["value"] > 0 and r["combined_render"]["value"] > 0
