#!/bin/bash
# driver-protocol A/B of the headline line: usage r3_drv.sh <outdir> ; prints ms_per_step of each run
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --only --cpu-budget 0 --no-pure-gather > $O/drv_$i.log 2>&1 || exit 1; done
for e in 2 4; do HSK_BENCH_EVERY=$e python bench.py --steps 20 --warmup 5 --only --cpu-budget 0 --no-pure-gather > $O/every_$e.log 2>&1 || exit 1; done
python bench.py --steps 200 --warmup 20 --only --cpu-budget 0 --no-pure-gather > $O/long.log 2>&1 || exit 1
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/*.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], round(d['ms_per_step']*1e3,1), 'noflush', round(d['ms_per_step_without_closing_flush']*1e3,1), 'flush', round(d['flush_us_in_timed_region'],1), 'fwd', round(d['roofline']['avg_us'],1), d['roofline']['launches'], d['lazy_sweep_cadence_steps'])
PY
