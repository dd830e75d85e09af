#!/bin/bash
# kernel trace of the materialised evaluation at one rank's shard width of the 8-GPU lfm2b evaluation (16384 x 16384)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_width; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/profiles/probes/eval_width.py 16384 > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
tail -1 $OUT/log.txt
python3 $ROOT/tools/kstats.py $OUT/trace 14
