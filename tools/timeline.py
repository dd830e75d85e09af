#!/usr/bin/env python3
"""Print the kernel timeline of the last few steps of a rocprofv3 --kernel-trace run.
usage: tools/timeline.py <dir> [anchor kernel prefix] [n_steps]"""
import csv
import glob
import sys

p = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
anchor = sys.argv[2] if len(sys.argv) > 2 else 'k_fwd_ugrad'
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = list(csv.DictReader(open(p)))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']),
             r['Kernel_Name'].split('(')[0].replace('void ', '')[:34], r.get('Queue_Id')) for r in rows)
idx = [i for i, k in enumerate(ks) if k[2].startswith(anchor)]
i0 = idx[-(n + 1)]
t0 = ks[i0][0]
for k in ks[i0:idx[-1] + 1]:
    print(f"{(k[0]-t0)/1e3:9.1f} {(k[1]-t0)/1e3:9.1f}  dur {(k[1]-k[0])/1e3:7.1f}  {k[2]:36s} q{k[3]}")
