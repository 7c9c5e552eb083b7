"""Per-kernel averages of the SQ / TCC counter passes of tools/collect_sq.sh -> <tag>_bench_pmc_sq.csv (one row per kernel, one column
per counter, plus a few derived ratios). SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over all waves;
SQ_INSTS_* count wave-instructions; SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE count LDS-array cycles (MI355X_MICROARCH.md)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    out, tag = sys.argv[1], sys.argv[2]
    table = defaultdict(dict)          # kernel -> counter -> (sum, n)
    for d in sorted(glob.glob(os.path.join(out, "sq_*"))):
        if not os.path.isdir(d):
            continue
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                s = table[row["Kernel_Name"]].setdefault(row["Counter_Name"], [0.0, 0])
                s[0] += float(row["Counter_Value"])
                s[1] += 1
    counters = sorted({c for k in table.values() for c in k})
    if not counters:
        print("no counter data found")
        return
    rows = []
    for k, cs in table.items():
        avg = {c: (cs[c][0] / cs[c][1] if c in cs else None) for c in counters}
        n = max(v[1] for v in cs.values())
        wc = avg.get("SQ_WAVE_CYCLES") or 0
        d = {
            "valu_insts_per_wave": (avg.get("SQ_INSTS_VALU") or 0) / avg["SQ_WAVES"] if avg.get("SQ_WAVES") else None,
            "lds_insts_per_wave": (avg.get("SQ_INSTS_LDS") or 0) / avg["SQ_WAVES"] if avg.get("SQ_WAVES") and avg.get("SQ_INSTS_LDS") is not None else None,
            "wait_any_share": (avg.get("SQ_WAIT_ANY") or 0) / wc if wc else None,
            "wait_inst_share": (avg.get("SQ_WAIT_INST_ANY") or 0) / wc if wc else None,
            "active_inst_share": (avg.get("SQ_ACTIVE_INST_ANY") or 0) / wc if wc else None,
            "lds_bank_conflict_share_of_lds_cycles": (avg.get("SQ_LDS_BANK_CONFLICT") or 0) / avg["SQ_LDS_IDX_ACTIVE"] if avg.get("SQ_LDS_IDX_ACTIVE") else None,
            "l2_hit_rate": (avg.get("TCC_HIT_sum") or 0) / ((avg.get("TCC_HIT_sum") or 0) + (avg.get("TCC_MISS_sum") or 0)) if (avg.get("TCC_HIT_sum") or avg.get("TCC_MISS_sum")) else None,
        }
        rows.append((k, n, avg, d))
    rows.sort(key=lambda r: -(r[2].get("SQ_WAVE_CYCLES") or 0))
    derived = ["valu_insts_per_wave", "lds_insts_per_wave", "wait_any_share", "wait_inst_share", "active_inst_share", "lds_bank_conflict_share_of_lds_cycles",
               "l2_hit_rate"]
    path = os.path.join(out, f"{tag}_bench_pmc_sq.csv")
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "dispatches"] + counters + derived)
        for k, n, avg, d in rows[:24]:
            w.writerow([k[:160], n] + [("" if avg[c] is None else round(avg[c])) for c in counters] + [("" if d[x] is None else round(d[x], 4)) for x in derived])
    print("wrote", path)
    for k, n, avg, d in rows[:8]:
        print(k[:60], {x: (None if d[x] is None else round(d[x], 3)) for x in derived})


if __name__ == "__main__":
    main()
