#!/bin/bash
# A/B of ENVIRONMENT settings of one library build on ONE box: tools/ab_env.sh "NAME=VAL ..." "NAME=VAL ..." ...  [REPS=2]  ("" = defaults)
# Each setting runs the headline step (no extras) REPS times, interleaved; prints ms/step and the per-op event timings.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
REPS=${REPS:-2}
for rep in $(seq $REPS); do
  for setting in "$@"; do
    env $setting python3 "$R/bench.py" --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('[$setting]', 'ms/step %.4f' % d['ms_per_step'], 'median %.4f' % d['step_ms']['median'], {n: k[n]['avg_ms'] for n in k if 'grid' in n})
"
  done
done
