#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_cfg5.py -x -q -m gpu > $O/test_cfg5.log 2>&1; echo "cfg5 tests rc=$?" | tee -a $O/test_cfg5.log
tail -5 $O/test_cfg5.log
timeout -k 10 600 python bench.py --workload cfg5 --steps 12 --warmup 4 --cpu-budget 0 > $O/bench_cfg5.log 2>&1; echo "bench cfg5 rc=$?"
tail -c 3000 $O/bench_cfg5.log
