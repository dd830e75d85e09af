#!/bin/bash
# A/B of two builds on one box (HSK_LIB_PATH): put the other build at hassaku_amd/libhsk_old_ab.so.
# usage: tools/ab_lib.sh <workload> <steps> <warmup> : alternates HSK_LIB_PATH old/new three times
W=$1; S=$2; U=$3
for i in 1 2 3; do
  for lib in libhsk_old_ab.so libhassaku_hip.so; do
    HSK_LIB_PATH=$PWD/hassaku_amd/$lib python bench.py --cpu-budget 0 --only --workload $W --steps $S --warmup $U 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('$W', '$lib', round(d['ms_per_step']*1e3,2), round(d['roofline']['avg_us'],2))" || exit 1
  done
done
