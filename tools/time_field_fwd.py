#!/usr/bin/env python3
"""The training forward of both networks at the headline batch (2 097 152 samples): foc_ffmlp_forward_planar + foc_color_head_forward against
foc_field_forward_train, bursts of back-to-back calls, median per call. FOCNERF_LIB_PATH selects another build."""
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch  # noqa: E402

from focnerf_amd._lib import lib, ptr, check, stream_of  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096 * 512
T, reps = 512, 30
g = torch.Generator(device="cuda").manual_seed(0)
planes = ((torch.rand(16, M, 2, device="cuda", generator=g) - 0.5)).half()
w_s = ((torch.rand(64 * (32 + 64 + 16), device="cuda", generator=g) - 0.5) * 0.4).half()
w_c = ((torch.rand(64 * (32 + 128 + 16), device="cuda", generator=g) - 0.5) * 0.4).half()
ray_sh = (torch.randn(M // T, 16, device="cuda", generator=g) * 0.3).half()
h = torch.empty(M, 16, dtype=torch.float16, device="cuda")
c = torch.empty(M, 4, dtype=torch.float16, device="cuda")
st = stream_of(h)


def two():
    check(lib.foc_ffmlp_forward_planar(ptr(planes), ptr(w_s), M, 32, 16, 64, 2, 0, 6, ptr(h), st), "s")
    check(lib.foc_color_head_forward(ptr(h), ptr(ray_sh), T, ptr(w_c), M, 64, 3, 0, ptr(c), 4, None, st), "c")


def one():
    check(lib.foc_field_forward_train(ptr(planes), ptr(w_s), 2, ptr(ray_sh), T, ptr(w_c), 3, 64, 0, M, ptr(h), ptr(c), 4, None, st), "f")


out = {"rows": M, "lib": os.path.basename(os.environ.get("FOCNERF_LIB_PATH", "libfocnerf_hip.so"))}
for name, fn in (("two_calls", two), ("fused", one), ("two_calls_again", two), ("fused_again", one)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(1e3 * s.elapsed_time(e) / reps)
    ts.sort()
    out[name + "_us"] = round(ts[2], 1)
print(json.dumps(out))
