#!/bin/bash
# kernel timelines of the headline step under different environments: usage r4_trace_env.sh "ENV=a" "ENV=b" ...
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for e in "$@"; do
  i=$((i+1))
  OUT=$ROOT/gpurun_out/prof_r4_env$i; rm -rf $OUT; mkdir -p $OUT
  env $e rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --cpu-budget 0 --only --no-pure-gather --steps 60 --warmup 10 > $OUT/bench_trace.log 2>&1 || { tail -5 $OUT/bench_trace.log; exit 1; }
  echo "== $e: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_trace.log | head -1)"
  python3 $ROOT/tools/timeline.py $OUT/trace k_fwd_part 2
done
