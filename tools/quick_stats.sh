#!/bin/bash
# rocprofv3 kernel stats of the headline step, top kernels only (run on the GPU box): tools/quick_stats.sh <tag>
TAG=${1:-q}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/quick_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-extras > "$OUT/stats.log" 2>&1
F=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
python3 - "$F" <<'PY' | tee "$OUT/summary.txt"
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):5d} avg_us {float(r["AverageNs"])/1e3:9.1f} pct {float(r["Percentage"]):5.1f}')
PY
tail -1 "$OUT/stats.log" | cut -c1-300
rm -rf "$OUT/stats"
