#!/usr/bin/env python3
"""Print the per-kernel summary of a rocprofv3 --kernel-trace --stats run.  usage: tools/kstats.py <dir> [n]"""
import csv
import glob
import sys

p = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for r in list(csv.DictReader(open(p)))[:n]:
    print(f"{r['Name'].split('(')[0][:90]:90s} calls {r['Calls']:>6s}  avg_us {float(r['AverageNs'])/1e3:9.2f}  {r['Percentage']:>6s}%")
