#!/bin/bash
# rocprofv3 kernel averages of tools/time_occ_ops.py for one library build and one batch size (run on the GPU box):
#   tools/prof_ops.sh <label> <lib.so | ""> <rays> [kernel-name filter regex]
# Environment switches set by the caller are inherited. Prints "label rays kernel calls avg_us" lines.
LABEL=$1; LIB=$2; RAYS=$3; FILT=${4:-"k_gbin|k_grid_fwd|k_mlp"}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_ops_$LABEL
mkdir -p "$OUT"
export RAYS REPS=${REPS:-10}
if [ -n "$LIB" ]; then export FOCNERF_LIB_PATH=$(realpath "$LIB"); fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/tools/time_occ_ops.py" > "$OUT/log.txt" 2>&1
F=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
python3 - "$F" "$LABEL" "$RAYS" "$FILT" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: r["Name"]):
    if re.search(sys.argv[4], r["Name"]):
        name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")[:60]
        print(f'{sys.argv[2]:12s} rays {sys.argv[3]:>6s} {name:60s} calls {int(r["Calls"]):4d} avg_us {float(r["AverageNs"])/1e3:8.1f}')
PY
rm -rf "$OUT"
