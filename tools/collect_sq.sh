#!/bin/bash
# SQ / LDS / atomic counters of the headline step's kernels (run on the GPU box): separate rocprofv3 --pmc passes (8 SQ slots per
# pass; never combined with a trace domain), summarised per kernel into gpurun_out/profiles/<tag>_bench_pmc_sq.csv.
# usage: tools/collect_sq.sh r02            (SQ_SCRIPT=tools/time_render_fixed.py tools/collect_sq.sh r02_render: another program's kernels)
set -o pipefail
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/profiles
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH_ARGS="--steps 6 --warmup 3 --no-cpu-baseline --no-extras"
PROG="$R/bench.py"
if [ -n "$SQ_SCRIPT" ]; then PROG="$R/$SQ_SCRIPT"; BENCH_ARGS=""; fi
pass() {  # name counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/sq_$name" -- python3 "$PROG" $BENCH_ARGS > "$OUT/sq_$name.log" 2>&1 || echo "pass $name failed (see $OUT/sq_$name.log)"
}
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
pass b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass c SQ_WAIT_INST_LDS SQ_INSTS_GDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_ATOMIC_RETURN SQ_THREAD_CYCLES_VALU
pass d TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
python3 "$R/tools/summarize_sq.py" "$OUT" "$TAG"
rm -rf "$OUT"/sq_a "$OUT"/sq_b "$OUT"/sq_c "$OUT"/sq_d      # raw counter dumps stay on the box (gpurun copies back at most 64 MiB)
