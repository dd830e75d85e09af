#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "presplit or fused or eval or topk" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for i in 1 2; do
 HSK_EVAL_WIDE=0 python bench.py --eval-only ml10m > $O/ml10m_old_$i.log 2>&1 || exit 1
 python bench.py --eval-only ml10m > $O/ml10m_new_$i.log 2>&1 || exit 1
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/ml10m*.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])['eval']['ml10m']
    print(f.split('/')[-1], round(d['users_per_s']/1e6,2), 'M users/s', round(d['tflops_fp32'],1), 'TF', d['ndcg@10_check'])
PY
