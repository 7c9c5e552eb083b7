"""Summarise the rocprofv3 runs of tools/collect_profiles.sh into the two CSVs kept under profiles/.

  <tag>_bench_kernel_stats.csv : rocprofv3's own kernel_stats.csv (name, calls, total/avg/min/max ns, percentage)
  <tag>_bench_pmc_hbm.csv      : per kernel, HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE
        FETCH_SIZE is reported in KB and, on gfx950, counts 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section),
        hence the x2; WRITE_SIZE (KB) is taken as is. Counters come from two separate --pmc passes.
"""
import csv
import glob
import os
import shutil
import sys
from collections import defaultdict


def newest(pattern):
    files = glob.glob(pattern, recursive=True)
    if not files:
        raise SystemExit(f"no file matches {pattern}")
    return max(files, key=os.path.getmtime)


def counter_avg(dirname, counter):
    path = newest(os.path.join(dirname, "**", "*counter_collection.csv"))
    acc = defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        a = acc[row["Kernel_Name"]]
        a[0] += float(row["Counter_Value"])
        a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def pmc_hbm_csv(fetch_dir, write_dir, dst):
    fetch = counter_avg(fetch_dir, "FETCH_SIZE")
    write = counter_avg(write_dir, "WRITE_SIZE")
    rows = []
    for k in sorted(set(fetch) | set(write)):
        f_kb, n = fetch.get(k, (0.0, 0))
        w_kb, nw = write.get(k, (0.0, 0))
        fb, wb = 2.0 * f_kb * 1024.0, w_kb * 1024.0
        rows.append((k, max(n, nw), round(f_kb), round(fb), round(wb), round(fb + wb)))
    rows.sort(key=lambda r: -r[5])
    with open(dst, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "dispatches", "FETCH_SIZE_KB_raw_avg", "fetch_bytes_corrected_x2", "write_bytes", "hbm_bytes_per_launch"])
        w.writerows(rows)


def main():
    if sys.argv[1] == "--pmc-only":          # summarize_profiles.py --pmc-only <fetch dir> <write dir> <dst csv>
        pmc_hbm_csv(sys.argv[2], sys.argv[3], sys.argv[4])
        print("wrote", sys.argv[4])
        return
    out, tag = sys.argv[1], sys.argv[2]
    stats = newest(os.path.join(out, "stats", "**", "*kernel_stats.csv"))
    shutil.copy(stats, os.path.join(out, f"{tag}_bench_kernel_stats.csv"))
    fetch = counter_avg(os.path.join(out, "pmc_fetch"), "FETCH_SIZE")
    write = counter_avg(os.path.join(out, "pmc_write"), "WRITE_SIZE")
    rows = []
    for k in sorted(set(fetch) | set(write)):
        f_kb, n = fetch.get(k, (0.0, 0))
        w_kb, nw = write.get(k, (0.0, 0))
        fb, wb = 2.0 * f_kb * 1024.0, w_kb * 1024.0
        rows.append((k, max(n, nw), round(f_kb), round(fb), round(wb), round(fb + wb)))
    rows.sort(key=lambda r: -r[5])
    with open(os.path.join(out, f"{tag}_bench_pmc_hbm.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "dispatches", "FETCH_SIZE_KB_raw_avg", "fetch_bytes_corrected_x2", "write_bytes", "hbm_bytes_per_launch"])
        w.writerows(rows)
    # matrix-core utilisation of the kernels that use it: SQ_VALU_MFMA_BUSY_CYCLES counts pipe cycles summed over the SIMDs
    # (32 per v_mfma_f32_32x32x16_f16, MI355X_MICROARCH.md); GRBM_GUI_ACTIVE the busy cycles summed over the 8 XCDs (observed: 8 x duration x
    # clock); utilisation = busy / (gui_active / 8 * 1024 SIMDs)
    mdir = os.path.join(out, "pmc_mfma")
    if os.path.isdir(mdir):
        busy = counter_avg(mdir, "SQ_VALU_MFMA_BUSY_CYCLES")
        mops = counter_avg(mdir, "SQ_INSTS_VALU_MFMA_MOPS_F16")
        act = counter_avg(mdir, "GRBM_GUI_ACTIVE")
        with open(os.path.join(out, f"{tag}_bench_pmc_mfma.csv"), "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["kernel", "dispatches", "mfma_busy_cycles_avg", "mfma_mops_f16_avg", "gui_active_cycles_avg", "mfma_pipe_utilisation"])
            for k in sorted(busy, key=lambda k: -busy[k][0]):
                b, n = busy[k]
                if b <= 0:
                    continue
                a = act.get(k, (0.0, 0))[0]
                w.writerow([k, n, round(b), round(mops.get(k, (0.0, 0))[0]), round(a), round(b / (a / 8.0 * 1024.0), 4) if a else ""])
    print("wrote", out, tag)


if __name__ == "__main__":
    main()
