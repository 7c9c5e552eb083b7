#!/bin/bash
# rocprofv3 kernel stats of the fixed-step render of one 800x800 view, top kernels (run on the GPU box): tools/quick_render_stats.sh <tag> [script]
TAG=${1:-q}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SCRIPT=${2:-$R/tools/time_render_fixed.py}
OUT=$R/gpurun_out/quickr_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$SCRIPT" > "$OUT/stats.log" 2>&1
F=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
python3 - "$F" <<'PY' | tee "$OUT/summary.txt"
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:12]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):5d} avg_us {float(r["AverageNs"])/1e3:9.1f} pct {float(r["Percentage"]):5.1f}')
PY
tail -1 "$OUT/stats.log" | cut -c1-300
rm -rf "$OUT/stats"
