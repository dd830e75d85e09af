#!/bin/bash
# evaluation chunk size (users per materialised score block) at the ml10m shape
set -o pipefail
cd $GRAFT_REPO_ROOT
for c in "$@"; do
  HSK_BENCH_EVAL_CHUNK=$c python bench.py --eval-only ml10m 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])['eval']['ml10m']; print('chunk $c', round(d['users_per_s']/1e6,2), 'M users/s', d['ndcg@10_check'])" || exit 1
done
