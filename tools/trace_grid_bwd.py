"""Prints start / duration of the grid-backward kernels of the last training step found in a rocprofv3 kernel trace (argv[1] = directory)."""
import csv, glob, os, sys
f = max(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_fs_sample" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = None
for r in rows[a:b]:
    if "k_gbin" in r["Kernel_Name"]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        t0 = t0 or s
        print(f"{(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f}  ({(e - s) / 1e3:6.1f} us)  {r['Kernel_Name'][:60]}  grid {r.get('Grid_Size_X', r.get('Grid_Size'))}")
