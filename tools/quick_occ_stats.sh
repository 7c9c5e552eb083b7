#!/bin/bash
# rocprofv3 kernel stats of the occupancy-grid training step (tools/prof_occupancy.py), per step: tools/quick_occ_stats.sh <tag>
TAG=${1:-q}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/occ_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/tools/prof_occupancy.py" > "$OUT/log.txt" 2>&1
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
python3 "$R/tools/occ_step_summary.py" "$(find "$OUT/stats" -name '*kernel_trace.csv' | head -1)" | tee "$OUT/summary.txt"
grep "ms/step" "$OUT/log.txt"
rm -rf "$OUT/stats"
