#!/bin/bash
# A/B recipes for one GPU box (run through gpurun; every recipe alternates its arms three times on the same box,
# because two boxes differ by more than most changes: +-2 us per step, up to 12 % on MFMA-dense loops).
#
#   tools/ab.sh env   "<bench args>" "ENV=a" "ENV=b" ...    environment switches (DESIGN.md "Switches"), headline step
#   tools/ab.sh lib   "<bench args>" <other.so>             two builds: hassaku_amd/<other.so> against libhassaku_hip.so
#   tools/ab.sh eval  <shape> "ENV=a" "ENV=b" ...           evaluation leg (ml10m | lfm2b): users/s
#   tools/ab.sh stages "<bench args>" "ENV=a" ...           every stage event-timed (perturbs the step; for shares only)
# examples:
#   tools/ab.sh env "--steps 200 --warmup 20" HSK_PIPE=0 HSK_PIPE=1
#   tools/ab.sh env "--steps 20 --warmup 5" HSK_SIDE_PRIO=0 HSK_SIDE_PRIO=-1          (the driver's protocol)
#   tools/ab.sh lib "--workload ml1m --steps 1920 --warmup 192" libhsk_old_ab.so
#   tools/ab.sh eval lfm2b HSK_FUSED_WGS=512 HSK_FUSED_WGS=1024
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
MODE=$1; shift
line() {  # prints one result line from bench.py's JSON on stdin: $1 = label, $2 = python expression over d
  python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('$1', $2)"
}
case $MODE in
  env)
    ARGS=$1; shift
    for i in 1 2 3; do for e in "$@"; do
      env $e python bench.py --cpu-budget 0 --only --no-pure-gather $ARGS 2>/dev/null |
        line "$e" "round(d['ms_per_step']*1e3,2), 'us/step, fwd', round(d['roofline']['avg_us'],2)" || exit 1
    done; done ;;
  lib)
    ARGS=$1; OTHER=$2
    for i in 1 2 3; do for lib in $OTHER libhassaku_hip.so; do
      HSK_LIB_PATH=$PWD/hassaku_amd/$lib python bench.py --cpu-budget 0 --only --no-pure-gather $ARGS 2>/dev/null |
        line "$lib" "round(d['ms_per_step']*1e3,2), 'us/step, fwd', round(d['roofline']['avg_us'],2)" || exit 1
    done; done ;;
  eval)
    SHAPE=$1; shift
    for i in 1 2 3; do for e in "$@"; do
      env $e python bench.py --eval-only $SHAPE 2>/dev/null |
        line "$e" "round(d['eval']['$SHAPE']['users_per_s']/1e6,3), 'M users/s', round(d['eval']['$SHAPE']['tflops_fp32'],1), 'TF', d['eval']['$SHAPE']['ndcg@10_check']" || exit 1
    done; done ;;
  stages)
    ARGS=$1; shift
    for e in "$@"; do
      env $e python bench.py --cpu-budget 0 --only --no-pure-gather --time-all-stages $ARGS 2>/dev/null |
        line "$e" "round(d['ms_per_step']*1e3,1), 'us/step', {k: round(v,1) for k,v in d['stage_us_per_step'].items()}" || exit 1
    done ;;
  *) echo "usage: see the head of tools/ab.sh"; exit 2 ;;
esac
