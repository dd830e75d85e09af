#!/bin/bash
# A/B of environment settings on one box, alternated three times.
# usage: tools/ab_env.sh "<bench args>" "ENV=a" "ENV=b" ...      e.g. tools/ab_env.sh "--steps 200 --warmup 20" HSK_PIPE=0 HSK_PIPE=1
set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
ARGS=$1; shift
for i in 1 2 3; do
  for e in "$@"; do
    env $e python bench.py --cpu-budget 0 --only --no-pure-gather $ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('$e', round(d['ms_per_step']*1e3,2), 'fwd', round(d['roofline']['avg_us'],2))" || exit 1
  done
done
