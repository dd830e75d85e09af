#!/usr/bin/env python3
"""Print the kernel timeline of one training step from a rocprofv3 --kernel-trace CSV (start/end in us relative
to the step's first kernel).  usage: tools_timeline.py <kernel_trace.csv> [step_index_from_end]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith(('k_prep_sample', 'k_prep_external'))]
i0, i1 = starts[-back - 1], starts[-back]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1]:
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]
    print(f'{s:8.1f} {e:8.1f} {e - s:7.1f}  q{r.get("Queue_Id", "?")}  {name}')
print('step span us:', (int(rows[i1]['Start_Timestamp']) - t0) / 1e3)
