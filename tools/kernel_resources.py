#!/usr/bin/env python3
"""tools/kernel_resources.py <file.hip> [name-filter]: per kernel VGPRs / AGPRs / scratch bytes per lane / occupancy / static LDS, from
hipcc -Rpass-analysis=kernel-resource-usage (compiles the file for gfx950, no output object)."""
import os
import re
import subprocess
import sys

csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "focnerf_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function",
       "-Wno-pass-failed", "-Rpass-analysis=kernel-resource-usage", "-c", sys.argv[1], "-o", "/dev/null"] + [a for a in sys.argv[3:]]
out = subprocess.run(cmd, cwd=csrc, capture_output=True, text=True).stderr
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows, cur = [], None
pats = {"v": r" VGPRs: (\d+)", "a": r"AGPRs: (\d+)", "scratch": r"ScratchSize \[bytes/lane\]: (\d+)", "occ": r"Occupancy \[waves/SIMD\]: (\d+)",
        "lds": r"LDS Size \[bytes/block\]: (\d+)"}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for k, p in pats.items():
        m = re.search(p, line)
        if m and cur is not None:
            cur[k] = m.group(1)
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    if flt in n:
        print(f"{n[:120]:120s} " + " ".join(f"{k}={r.get(k)}" for k in pats))
