#!/usr/bin/env python3
"""The two fused MLP backward calls of the headline step in isolation (2 097 152 rows; sigma network planar 32 -> 64 -> 64 -> 16 with planar input
gradients, colour head 32 -> 64 -> 64 -> 64 -> [M,4] with grad_h), event-timed, for A/B of library builds on one box:
    FOCNERF_LIB_PATH=_ab/lib_x.so python tools/time_mlp_bwd.py [rows] [reps]
Prints one JSON line: microseconds per call (median and min over reps) for both, kernel + slot reduce together."""
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch  # noqa: E402

from focnerf_amd._lib import lib, ptr, check, stream_of  # noqa: E402
from focnerf_amd.backend import _scratch  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096 * 512
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    T = 512
    dev = torch.device("cuda")
    torch.manual_seed(0)
    g = torch.Generator(device=dev).manual_seed(1)
    planes = (torch.rand(16, M, 2, device=dev, generator=g) - 0.5).half()
    w_s = ((torch.rand(64 * (32 + 64 + 16), device=dev, generator=g) - 0.5) * 0.4).half()
    w_c = ((torch.rand(64 * (32 + 128 + 16), device=dev, generator=g) - 0.5) * 0.4).half()
    grad_h = (torch.randn(M, 16, device=dev, generator=g) * 0.01).half()
    gplanes = torch.empty_like(planes)
    gw_s, gw_c = torch.empty_like(w_s), torch.empty_like(w_c)
    h = (torch.randn(M, 16, device=dev, generator=g) * 0.3).half()
    ray_sh = (torch.randn(M // T, 16, device=dev, generator=g) * 0.3).half()
    grad_c = (torch.randn(M, 4, device=dev, generator=g) * 0.01).half()
    grad_h0 = (torch.randn(M, device=dev, generator=g) * 0.01).half()
    gh = torch.empty_like(h)
    ws_s = _scratch.get("ffmlp_ws", lib.foc_ffmlp_backward_workspace_bytes(32, 64, 2), dev)
    ws_c = _scratch.get("ffmlp_ws2", lib.foc_ffmlp_backward_workspace_bytes(32, 64, 3), dev)
    st = stream_of(h)

    def sigma():
        check(lib.foc_ffmlp_backward_planar(ptr(grad_h), ptr(planes), ptr(w_s), M, 32, 16, 64, 2, 0, 6, 1, ptr(gplanes), ptr(gw_s), ptr(ws_s), ws_s.numel(), st), "sigma bwd")

    def colour():
        check(lib.foc_color_head_backward(ptr(grad_c), ptr(h), ptr(ray_sh), T, ptr(grad_h0), ptr(w_c), M, 64, 3, 0, ptr(gh), ptr(gw_c), ptr(ws_c), ws_c.numel(), 4, None, None, st),
              "colour bwd")
    out = {"rows": M, "lib": os.path.basename(os.environ.get("FOCNERF_LIB_PATH", "libfocnerf_hip.so"))}
    for name, fn in (("sigma", sigma), ("colour", colour)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):                                   # five bursts of `reps` back-to-back calls (no idle GPU between calls, as inside a step)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                fn()
            e.record()
            torch.cuda.synchronize()
            ts.append(1e3 * s.elapsed_time(e) / reps)
        ts.sort()
        out[name + "_us"] = {"median": round(ts[len(ts) // 2], 1), "min": round(ts[0], 1)}
    out["sum_median_us"] = round(out["sigma_us"]["median"] + out["colour_us"]["median"], 1)
    out["checksum"] = [float(gw_s.float().abs().sum()), float(gw_c.float().abs().sum()), float(gplanes.float().abs().sum()), float(gh.float().abs().sum())]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
