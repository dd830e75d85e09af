#!/bin/bash
# configs[4], one rank's share: dense against lazy AdamW on the item shard (73 % of its rows are touched per step)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/cfg5lazy; mkdir -p $O
for v in 1 0; do
  HSK_CFG5_LAZY_ITEMS=$v timeout -k 10 500 python bench.py --workload cfg5 --steps 12 --warmup 4 --cpu-budget 0 > $O/lazy$v.log 2>&1 || { tail -5 $O/lazy$v.log; exit 1; }
  python - <<PY
import json
d=json.loads([l for l in open('$O/lazy$v.log') if l.startswith('{')][-1])['workloads']['cfg5_shard']
print('lazy_items=$v', 'ms/step', round(d['ms_per_step'],3), 'with sweep', round(d['ms_per_step_with_amortised_sweep'],3), {k: round(x,1) for k,x in d['stage_us_per_step'].items()}, d['lazy_sweep'])
PY
done
