#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/profile.sh into profiles/<name>/{kernel_stats.csv, pmc_summary.json}.
usage: tools/pmc_summary.py gpurun_out/prof_<tag> profiles/<name>
Only this repo's kernels (k_*) are kept; counter values are the per-dispatch means, in the units rocprofv3 reports."""
import csv
import glob
import json
import os
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
csv.field_size_limit(1 << 30)


def short(name):
    m = re.match(r'_Z(\d+)', name)       # names the profiler could not demangle (a __bf16 parameter): take the identifier
    if m:
        return name[m.end():m.end() + int(m.group(1))]
    return name.split('(')[0].replace('void ', '').strip()


summary = {}
for counter, sub in (('FETCH_SIZE', 'pmc_fetch'), ('WRITE_SIZE', 'pmc_write')):
    for path in glob.glob(os.path.join(src, sub, '**', '*counter_collection.csv'), recursive=True):
        acc = {}
        for r in csv.DictReader(open(path)):
            if r['Counter_Name'] != counter:
                continue
            k = short(r['Kernel_Name'])
            if not k.startswith('k_'):
                continue
            a = acc.setdefault(k, [0.0, 0])
            a[0] += float(r['Counter_Value'])
            a[1] += 1
        for k, (tot, n) in acc.items():
            row = summary.setdefault(k, {})
            row[counter + '_KB_mean'] = tot / n
            row['dispatches'] = n
# any further counters of an extra pass (tools/profile.sh PMC_EXTRA=...): per-dispatch means under their own names
for path in glob.glob(os.path.join(src, 'pmc_extra', '**', '*counter_collection.csv'), recursive=True):
    acc = {}
    for r in csv.DictReader(open(path)):
        k = short(r['Kernel_Name'])
        if not k.startswith('k_'):
            continue
        a = acc.setdefault((k, r['Counter_Name']), [0.0, 0])
        a[0] += float(r['Counter_Value'])
        a[1] += 1
    for (k, c), (tot, n) in acc.items():
        summary.setdefault(k, {})[c + '_mean'] = tot / n
for k, row in summary.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES_mean' in row and row.get('SQ_BUSY_CYCLES_mean'):
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD-ish unit, SQ_BUSY_CYCLES per SQ (MI355X guide: quote the ratio,
        # not an absolute): the share of the kernel's busy time in which a matrix pipe was busy
        if row['SQ_VALU_MFMA_BUSY_CYCLES_mean'] % float(1 << 32) == 0.0 and row['SQ_VALU_MFMA_BUSY_CYCLES_mean'] > 0:
            row['mfma_busy_over_sq_busy'] = None      # the counter read a whole multiple of 2^32: it wrapped or saturated
            row['note'] = 'SQ_VALU_MFMA_BUSY_CYCLES saturated over this dispatch; no ratio quoted'
        else:
            row['mfma_busy_over_sq_busy'] = row['SQ_VALU_MFMA_BUSY_CYCLES_mean'] / row['SQ_BUSY_CYCLES_mean']
json.dump(summary, open(os.path.join(dst, 'pmc_summary.json'), 'w'), indent=1)

for path in glob.glob(os.path.join(src, 'trace', '**', '*kernel_stats.csv'), recursive=True):
    rows = list(csv.DictReader(open(path)))
    with open(os.path.join(dst, 'kernel_stats.csv'), 'w', newline='') as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        for r in rows:
            r['Name'] = short(r['Name'])[:80]
            w.writerow(r)
print(json.dumps({k: v for k, v in summary.items() if k.startswith(('k_fwd', 'k_item', 'k_score', 'k_topk'))}, indent=1))
