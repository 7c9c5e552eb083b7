#!/bin/bash
# A/B of ENVIRONMENT settings on the fixed-step render (800x800 view) on ONE box: tools/ab_env_render.sh "NAME=VAL ..." ...   ("" = defaults)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2; do
  for setting in "$@"; do
    echo -n "[$setting] "; env $setting FIELDS=0 python3 "$R/tools/time_render_fixed.py" 2>/dev/null | tail -1
  done
done
