#!/bin/bash
# A/B of the fused MLP backward forms on one box (run on the GPU box): tests with FOC_MLP_BWD_PRIV = 1 and 2, then rocprofv3 kernel stats of
# the headline step for 0 / 1 / 2.   tools/ab_mlp_priv.sh <tag> [notest]
TAG=${1:-p}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/abpriv_$TAG
mkdir -p "$OUT"
cd "$R"
if [ "$2" != "notest" ]; then
  for v in 1 2; do
    FOC_MLP_BWD_PRIV=$v timeout -k 10 900 python3 -m pytest tests/test_gpu_ffmlp.py tests/test_gpu_network.py tests/test_gpu_network_foc.py tests/test_gpu_fixedstep.py tests/test_gpu_occtrain.py -x -q > "$OUT/pytest_$v.log" 2>&1
    rc=$?
    tail -3 "$OUT/pytest_$v.log"
    if [ $rc -ne 0 ]; then echo "tests failed with FOC_MLP_BWD_PRIV=$v (rc $rc)"; exit 1; fi
  done
fi
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2; do
  export FOC_MLP_BWD_PRIV=$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$v" -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-extras > "$OUT/stats_$v.log" 2>&1 || { echo "bench failed for $v"; tail -5 "$OUT/stats_$v.log"; exit 1; }
  F=$(find "$OUT/stats_$v" -name '*kernel_stats.csv' | head -1)
  echo "== FOC_MLP_BWD_PRIV=$v"
  python3 - "$F" <<'PY' | tee "$OUT/summary_$v.txt"
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:16]:
    if "mlp" in r["Name"] or "grid" in r["Name"] or "gbin" in r["Name"]:
        print(f'{r["Name"][:86]:86s} calls {int(r["Calls"]):5d} avg_us {float(r["AverageNs"])/1e3:9.1f} pct {float(r["Percentage"]):5.1f}')
PY
  tail -1 "$OUT/stats_$v.log" | cut -c1-160
  rm -rf "$OUT/stats_$v"
done
