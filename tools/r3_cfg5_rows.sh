#!/bin/bash
# configs[4], one rank's share: D-sliced against whole-row item pass
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/cfg5rows; mkdir -p $O
for v in 1 0 1 0; do
  HSK_ITEM_ROWS_BIG=$v timeout -k 10 500 python bench.py --workload cfg5 --steps 12 --warmup 4 --cpu-budget 0 > $O/rows$v.log 2>&1 || { tail -5 $O/rows$v.log; exit 1; }
  python - <<PY
import json
d=json.loads([l for l in open('$O/rows$v.log') if l.startswith('{')][-1])['workloads']['cfg5_shard']
print('rows=$v', 'ms/step', round(d['ms_per_step'],3), {k: round(x,1) for k,x in d['stage_us_per_step'].items()}, 'loss', d['loss_last_step_local_share'])
PY
done
