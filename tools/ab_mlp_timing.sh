#!/bin/bash
# A/B of library builds (tools/build_variant.sh ... -DFOC_TIMING_*: some give wrong results, time only) against the shipped library, one box:
# the two backward calls of the headline step in isolation (tools/time_mlp_bwd.py).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for rep in 1 2 3; do
  python3 tools/time_mlp_bwd.py 2>/dev/null
  for lib in _ab/lib_t_*.so; do
    FOCNERF_LIB_PATH=$(realpath "$lib") python3 tools/time_mlp_bwd.py 2>/dev/null
  done
done
