#!/bin/bash
# Collects the rocprofv3 evidence bench.py's `roofline` refers to (run on the GPU box via gpurun):
#   1. --kernel-trace --stats of the default bench command (kernel durations),
#   2. separate --pmc passes (FETCH_SIZE; WRITE_SIZE; MFMA busy cycles) — never combined with a trace domain,
# and summarises them into profiles/<tag>_bench_kernel_stats.csv and profiles/<tag>_bench_pmc_hbm.csv
# (written under gpurun_out/profiles/, copy them into profiles/ afterwards).
# usage: tools/collect_profiles.sh r01
set -e -o pipefail
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/profiles
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH_ARGS="--steps 8 --warmup 4 --no-cpu-baseline --no-extras"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/bench.py" $BENCH_ARGS > "$OUT/stats.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$R/bench.py" $BENCH_ARGS > "$OUT/pmc_fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$R/bench.py" $BENCH_ARGS > "$OUT/pmc_write.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma" -- python3 "$R/bench.py" $BENCH_ARGS > "$OUT/pmc_mfma.log" 2>&1
python3 "$R/tools/summarize_profiles.py" "$OUT" "$TAG"
# kernel-trace summaries of the three other measured paths (configs[2] training step, both render loops)
for pair in occupancy_train:prof_occupancy.py render_fixed:time_render_fixed.py render_occupancy:time_render_occ.py; do   # (render_occupancy: the native loop, one C call per iteration)
  name=${pair%%:*}; script=${pair##*:}
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -- python3 "$R/tools/$script" > "$OUT/$name.log" 2>&1
  cp "$(find "$OUT/$name" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_${name}_kernel_stats.csv"
done
# HBM counters of the configs[2] training step (two separate --pmc passes, as for the bench)
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/occ_pmc_fetch" -- python3 "$R/tools/prof_occupancy.py" > "$OUT/occ_pmc_fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/occ_pmc_write" -- python3 "$R/tools/prof_occupancy.py" > "$OUT/occ_pmc_write.log" 2>&1
python3 "$R/tools/summarize_profiles.py" --pmc-only "$OUT/occ_pmc_fetch" "$OUT/occ_pmc_write" "$OUT/${TAG}_occupancy_train_pmc_hbm.csv"
# the fixed-step render once more with every chunk on ONE stream (with the default two streams the kernels of neighbouring chunks overlap
# and rocprofv3's durations are those of kernels sharing the chip): the kernel stats and the HBM counters bench.py's roofline.render
# refers to. The environment is exported here — nothing stands between `--` and python3.
export FOC_RENDER_STREAMS=1 VIEWS=4 FIELDS=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/render_fixed_1s" -- python3 "$R/tools/time_render_fixed.py" > "$OUT/render_fixed_1s.log" 2>&1
cp "$(find "$OUT/render_fixed_1s" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_render_fixed_kernel_stats.csv"
export VIEWS=1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/render_pmc_fetch" -- python3 "$R/tools/time_render_fixed.py" > "$OUT/render_pmc_fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/render_pmc_write" -- python3 "$R/tools/time_render_fixed.py" > "$OUT/render_pmc_write.log" 2>&1
python3 "$R/tools/summarize_profiles.py" --pmc-only "$OUT/render_pmc_fetch" "$OUT/render_pmc_write" "$OUT/${TAG}_render_fixed_pmc_hbm.csv"
unset FOC_RENDER_STREAMS VIEWS FIELDS
# the raw traces (hundreds of MB) stay on the box: gpurun copies back at most 64 MiB of gpurun_out/
rm -rf "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_mfma" "$OUT/occupancy_train" "$OUT/render_fixed" "$OUT/render_occupancy" \
       "$OUT/render_fixed_1s" "$OUT/render_pmc_fetch" "$OUT/render_pmc_write" "$OUT/occ_pmc_fetch" "$OUT/occ_pmc_write"
