"""tools/rccl_overlap_from_trace.py <kernel_trace.csv>: from a rocprofv3 --kernel-trace CSV of `ONLY=overlap tools/time_rccl_one_rank.py`, how much of
the time RCCL's kernels run do kernels of this library run AT THE SAME TIME (interval intersection of start / end timestamps)? One line of JSON.
Two streams on one hardware queue never overlap: the intersection is then zero whatever the host enqueued 'asynchronously'."""
import csv, json, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
is_rccl = lambda n: re.search(r"nccl|rccl", n, re.I) is not None
is_lib = lambda n: re.search(r"\bk_[a-z]|_Z\d+k_", n) is not None
rc = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if is_rccl(r["Kernel_Name"]))
lb = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if is_lib(r["Kernel_Name"]))
# the big collectives only (the all-to-all of a 134 MB piece), not barriers / small gathers
big = [iv for iv in rc if iv[1] - iv[0] > 20000]
merged = []
for a, b in lb:                                          # union of the library's kernel intervals
    if merged and a <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], b)
    else:
        merged.append([a, b])
inter = 0
j = 0
for a, b in big:
    while j < len(merged) and merged[j][1] <= a:
        j += 1
    k = j
    while k < len(merged) and merged[k][0] < b:
        inter += max(0, min(b, merged[k][1]) - max(a, merged[k][0]))
        k += 1
tot = sum(b - a for a, b in big)
names = sorted({r["Kernel_Name"][:60] for r in rows if is_rccl(r["Kernel_Name"])})
print(json.dumps({"rccl_kernels": len(rc), "large_collectives": len(big), "rccl_time_ms": tot / 1e6, "avg_us_per_large_collective": tot / max(1, len(big)) / 1e3,
                  "concurrent_with_library_kernels_ms": inter / 1e6, "concurrent_fraction": inter / tot if tot else None, "rccl_kernel_names": names}))
