#!/bin/bash
# top-k over materialised rows: tests, then the ml10m evaluation pass (and the 8-GPU shard width probe) on the new build
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_dist.py -x -q -m gpu -k "presplit or fused or eval or topk or metrics" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
for i in 1 2; do
 python bench.py --eval-only ml10m > $O/ml10m_$i.log 2>&1 || exit 1
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/ml10m*.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])['eval']['ml10m']
    print(f.split('/')[-1], round(d['users_per_s']/1e6,2), 'M users/s', round(d['tflops_fp32'],1), 'TF', d['ndcg@10_check'])
PY
timeout -k 10 300 python profiles/probes/eval_width.py 16384 10677 > $O/width.log 2>&1; tail -4 $O/width.log
