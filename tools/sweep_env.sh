#!/bin/bash
# tools/sweep_env.sh VAR v1 v2 ... : headline ms/step and grid-backward op time for each value of an environment knob (GPU box)
VAR=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for v in "$@"; do
  env $VAR=$v python3 "$R/bench.py" --steps 30 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('$VAR=$v', 'ms/step %.4f' % d['ms_per_step'], 'median %.4f' % d['step_ms']['median'], {n: k[n]['avg_ms'] for n in k})
"
done
