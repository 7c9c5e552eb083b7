#!/bin/bash
# A/B helper: k_grid_fwd_counted duration in the occupancy-grid training step for several FOC_GRID_FUSE_SMALL settings.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export FOC_GRID_FUSE_SMALL=$v
  rm -rf /tmp/occ_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/occ_$v -- python3 "$R/tools/prof_occupancy.py" > /tmp/occ_$v.log 2>&1 || exit 1
  f=$(find /tmp/occ_$v -name '*kernel_stats.csv' | head -1)
  echo "fuse=$v $(grep 'ms/step' /tmp/occ_$v.log)"
  python3 -c "import csv,sys; [print('   ', r['Name'].split('(')[0][-40:], r['Calls'], round(float(r['AverageNs'])/1000,1), 'us') for r in csv.DictReader(open(sys.argv[1])) if 'k_grid_fwd' in r['Name'] or 'k_gbin' in r['Name']]" "$f"
done
