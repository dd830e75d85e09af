#!/bin/bash
# A/B helper: run bench.py with every stage timed under different env settings and print one line each.
# usage: [ABFLAGS="--workload ml1m --no-prefetch"] tools/ab.sh "NAME=VAL ..." "NAME=VAL ..." ...   ("" = defaults)
for cfg in "$@"; do
  out=$(env $cfg timeout -k 10 300 python bench.py --steps 128 --warmup 10 --cpu-budget 0 --time-all-stages $ABFLAGS 2>&1 | tail -1)
  echo "$out" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[$cfg]', round(d['ms_per_step']*1e3,1), 'us/step', {k: round(v,1) for k,v in d['stage_us_per_step'].items()})"
done
