#!/bin/bash
# Kernel timelines (rocprofv3 --kernel-trace) of one bench command under one or more environments.
# usage: tools/trace.sh <anchor kernel prefix> "<bench args>" "ENV=a" ["ENV=b" ...]
#   e.g. tools/trace.sh k_fwd_part "--only --no-pure-gather --steps 60 --warmup 10" HSK_PIPE=1 HSK_PIPE=0
#        tools/trace.sh k_score_topk "--eval-only lfm2b" HSK_EVAL_X3=1
# Prints ms_per_step (training legs) and the last two steps' timeline (tools/timeline.py) / the kernel averages
# (tools/kstats.py); the traces stay under gpurun_out/prof_trace<i>/.  The environment goes in FRONT of rocprofv3: the
# profiled program itself must follow `--` (an env / bash hop behind it would exec after the GPU is initialised).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
ANCHOR=$1; ARGS=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
i=0
for e in "$@"; do
  i=$((i+1))
  OUT=$ROOT/gpurun_out/prof_trace$i; rm -rf $OUT; mkdir -p $OUT
  env $e rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --cpu-budget 0 $ARGS > $OUT/bench_trace.log 2>&1 || { tail -5 $OUT/bench_trace.log; exit 1; }
  echo "== $e: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_trace.log | head -1)"
  python3 $ROOT/tools/timeline.py $OUT/trace $ANCHOR 2 || true
  python3 $ROOT/tools/kstats.py $OUT/trace 10 || true
done
