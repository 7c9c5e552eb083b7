"""Shared helpers for the GPU parity tests."""
import math

import numpy as np
import torch

from focnerf_amd import synthetic


def scene(bound, N, seed=0, H=64, W=64, dev="cuda", sigma0=50.0, radius=2.0):
    """Rays of a small synthetic view + the packed analytic occupancy grid (host copies included)."""
    cascade = 1 + math.ceil(math.log2(bound))
    grid = synthetic.analytic_density_grid(bound, sigma0=sigma0, device="cpu")
    thresh = min(float(grid.clamp(min=0).mean()), 10.0)
    bits = synthetic.packbits_host(grid, thresh)
    rays_o, rays_d = synthetic.make_view_rays(H, W, bound, 1, seed=seed, device="cpu", radius=radius)
    rays_o, rays_d = rays_o[0], rays_d[0]
    if N < rays_o.shape[0]:
        g = torch.Generator().manual_seed(seed + 100)
        sel = torch.randperm(rays_o.shape[0], generator=g)[:N]
        rays_o, rays_d = rays_o[sel].contiguous(), rays_d[sel].contiguous()
    aabb = torch.tensor([-bound, -bound, -bound, bound, bound, bound], dtype=torch.float32)
    return dict(bound=float(bound), cascade=cascade, grid=grid, thresh=thresh, bits=bits, rays_o=rays_o, rays_d=rays_d, aabb=aabb)


def to_np(t):
    return t.detach().cpu().numpy()


def half_ulp(x):
    """Spacing of fp16 at |x| (elementwise, numpy float32 in)."""
    ax = np.maximum(np.abs(x.astype(np.float32)), 2.0 ** -14)
    return (2.0 ** (np.floor(np.log2(ax)) - 10)).astype(np.float32)


def assert_half_close(got, want, ulps=2.0, atol=0.0, what=""):
    got = np.asarray(got, dtype=np.float32)
    want = np.asarray(want, dtype=np.float32)
    tol = ulps * half_ulp(want) + atol
    bad = np.abs(got - want) > tol
    assert not bad.any(), f"{what}: {bad.sum()} / {bad.size} elements off by more than {ulps} half-ulp (+{atol}); " \
                          f"worst |diff|={np.abs(got - want).max()}"


def assert_bits_equal(got, ref, what=""):
    """Bit-for-bit equality with a useful report (count, worst cases as values and hex)."""
    got = np.ascontiguousarray(got)
    ref = np.ascontiguousarray(ref)
    assert got.shape == ref.shape and got.dtype == ref.dtype, f"{what}: shape/dtype {got.shape}/{got.dtype} vs {ref.shape}/{ref.dtype}"
    it = {2: np.uint16, 4: np.uint32, 1: np.uint8, 8: np.uint64}[got.dtype.itemsize]
    gb, rb = got.view(it), ref.view(it)
    # +0 and -0 are distinct bit patterns but the same value; report them separately
    bad = gb != rb
    if not bad.any():
        return
    idx = np.argwhere(bad)
    lines = [f"{what}: {bad.sum()} / {bad.size} elements differ"]
    for k in idx[:12]:
        k = tuple(k)
        lines.append(f"  at {k}: got {got[k]!r} (0x{int(gb[k]):x})  want {ref[k]!r} (0x{int(rb[k]):x})")
    raise AssertionError("\n".join(lines))


def build_c_abi_consumer(tmp_path):
    """tests/c_abi_consumer.cpp -> an executable under tmp_path: a caller of include/focnerf.h written against the HIP runtime API only (plain g++,
    no torch, no Python), linked with the library and with the C oracle it checks the results against. Returns the executable's path."""
    import os
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    exe = os.path.join(str(tmp_path), "c_abi_consumer")
    libs = [os.path.join(repo, "focnerf_amd"), os.path.join(repo, "oracle", "_build"), os.path.join(rocm, "lib")]
    cmd = ["g++", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"), "-I" + os.path.join(repo, "include"),
           os.path.join(repo, "tests", "c_abi_consumer.cpp"), "-o", exe, "-lfocnerf_hip", "-loracle", "-lamdhip64"]
    for d in libs:
        cmd += ["-L" + d, "-Wl,-rpath," + d]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe
