#!/bin/bash
# A/B of library builds on the fixed-step render (800x800 view) on ONE box: tools/ab_render.sh libA.so libB.so ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2; do
  for lib in "$@"; do
    echo -n "$(basename $lib) "; FOCNERF_LIB_PATH=$(realpath "$lib") python3 "$R/tools/time_render_fixed.py" 2>/dev/null | tail -1
  done
done
