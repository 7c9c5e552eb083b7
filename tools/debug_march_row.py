"""Debug helper: one march_rays call, 16-lanes-per-ray form vs one-ray-per-lane form, progress lines to stdout (run under `timeout`)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import scene
from focnerf_amd.backend import _raymarching
from focnerf_amd import raymarching

N = int(os.environ.get("N", "2000"))
s = scene(2, N, seed=13)
o, d, bits = s["rays_o"].cuda(), s["rays_d"].cuda(), s["bits"].cuda()
nears, fars = raymarching.near_far_from_aabb(o, d, s["aabb"].cuda(), 0.2)
alive = torch.arange(N, dtype=torch.int32, device="cuda")
print("setup done", flush=True)
for n_step in (1, 3, 8):
    outs = {}
    for form, v in (("lane", 0), ("row", 1000000000)):
        from focnerf_amd import _lib
        _lib.set_option("FOC_MARCH_RAYS_ROW_MAX", v)
        M = N * n_step
        x = torch.zeros(M, 3, device="cuda"); dd = torch.zeros(M, 3, device="cuda"); dl = torch.zeros(M, 2, device="cuda")
        t0 = time.perf_counter()
        _raymarching.march_rays(N, n_step, alive, nears.clone(), o, d, s["bound"], 1 / 128, 1024, s["cascade"], 128, bits, nears, fars, x, dd, dl,
                                torch.zeros(N, device="cuda"))
        torch.cuda.synchronize()
        print(form, "n_step", n_step, "ms", 1000 * (time.perf_counter() - t0), flush=True)
        outs[form] = (x.cpu().numpy(), dd.cpu().numpy(), dl.cpu().numpy())
    for other in ("row",):
        for a, b, name in zip(outs["lane"], outs[other], ("xyzs", "dirs", "deltas")):
            same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
            print(" ", other, name, "bitwise equal" if same else f"DIFFER at {np.argwhere(a != b)[:5].tolist()}", flush=True)
