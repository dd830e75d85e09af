#!/bin/bash
# configs[4], one rank's share: dense against lazy AdamW on the item shard (73 % of its rows are touched per step);
# HSK_ITEM_ROWS_SHARD=1: the lazy item pass on whole 4 KB rows instead of four 1 KB slices
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/cfg5lazy; mkdir -p $O
for v in "1 1" "1 0" "0 0" "1 1" "0 0"; do
  set -- $v
  HSK_CFG5_LAZY_ITEMS=$1 HSK_ITEM_ROWS_SHARD=$2 timeout -k 10 500 python bench.py --workload cfg5 --steps 12 --warmup 4 --cpu-budget 0 > $O/lazy$1$2.log 2>&1 || { tail -5 $O/lazy$1$2.log; exit 1; }
  python - <<PY
import json
d=json.loads([l for l in open('$O/lazy$1$2.log') if l.startswith('{')][-1])['workloads']['cfg5_shard']
print('lazy_items=$1 rows=$2', 'ms/step', round(d['ms_per_step'],3), 'with sweep', round(d['ms_per_step_with_amortised_sweep'],3), {k: round(x,1) for k,x in d['stage_us_per_step'].items()})
PY
done
