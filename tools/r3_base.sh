#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a; mkdir -p $O
for i in 1 2; do python bench.py --steps 20 --warmup 5 --only --cpu-budget 0 > $O/drv_$i.log 2>&1 || exit 1; done
python bench.py --steps 200 --warmup 20 --only --cpu-budget 0 > $O/long.log 2>&1 || exit 1
python bench.py --steps 20 --warmup 5 --only --cpu-budget 0 --time-all-stages > $O/stages20.log 2>&1 || exit 1
python bench.py --steps 200 --warmup 20 --only --cpu-budget 0 --time-all-stages > $O/stages200.log 2>&1 || exit 1
python bench.py --steps 20 --warmup 5 --only --cpu-budget 0 --sharded > $O/shard20.log 2>&1 || exit 1
python bench.py --steps 200 --warmup 20 --only --cpu-budget 0 --sharded > $O/shard200.log 2>&1 || exit 1
grep -h -o '"ms_per_step": [0-9.]*' $O/*.log
