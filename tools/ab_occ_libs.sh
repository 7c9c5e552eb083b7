#!/bin/bash
# Same-box A/B of library builds on the configs[2] training step: rocprofv3 kernel stats per step of tools/prof_occupancy.py for the shipped
# library and for every lib given (FOCNERF_LIB_PATH), interleaved twice.   tools/ab_occ_libs.sh <tag> _ab/lib_prev.so ...
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/abocc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for lib in shipped "$@"; do
    name=$(basename "$lib" .so)
    if [ "$lib" = shipped ]; then unset FOCNERF_LIB_PATH; else export FOCNERF_LIB_PATH=$(realpath "$R/$lib"); fi
    rm -rf "$OUT/stats"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/tools/prof_occupancy.py" > "$OUT/log_${name}_$rep.txt" 2>&1
    echo "== $name (rep $rep)"
    python3 "$R/tools/occ_step_summary.py" "$(find "$OUT/stats" -name '*kernel_trace.csv' | head -1)" | head -18 | tee "$OUT/summary_${name}_$rep.txt" | head -2
    grep "ms/step" "$OUT/log_${name}_$rep.txt"
  done
done
rm -rf "$OUT/stats"
