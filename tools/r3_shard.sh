#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_dist.py tests/test_cfg5.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
python bench.py --steps 200 --warmup 20 --only --cpu-budget 0 --sharded > $O/shard200.log 2>&1 || { tail -5 $O/shard200.log; exit 1; }
HSK_SHARD_NATIVE=0 python bench.py --steps 200 --warmup 20 --only --cpu-budget 0 --sharded > $O/shard200_phased.log 2>&1 || exit 1
python bench.py --steps 20 --warmup 5 --only --cpu-budget 0 --sharded > $O/shard20.log 2>&1 || exit 1
grep -h -o '"ms_per_step": [0-9.]*' $O/shard200.log $O/shard200_phased.log $O/shard20.log
