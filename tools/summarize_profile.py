#!/usr/bin/env python
"""Condense rocprofv3 output of tools/profile_round.sh into a small markdown summary (profiles/rNN_summary.md)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]


def short(name):
    for key, lab in (('igemm_kernel<128, false>', 'igemm_n128'), ('igemm_kernel<64, false>', 'igemm_n64'),
                     ('igemm_kernel<64, true>', 'igemm_c4_n64'), ('wgrad_kernel<128>', 'wgrad_m128'),
                     ('wgrad_kernel<64>', 'wgrad_m64'), ('igemm_kernelILi128ELb0', 'igemm_n128'),
                     ('igemm_kernelILi64ELb0', 'igemm_n64'), ('igemm_kernelILi64ELb1', 'igemm_c4_n64'),
                     ('wgrad_kernelILi128', 'wgrad_m128'), ('wgrad_kernelILi64', 'wgrad_m64')):
        if key in name:
            return lab
    name = name.replace('void ', '').replace('(anonymous namespace)::', '').replace('_ZN12_GLOBAL__N_1', '')
    return name.split('(')[0][:48]


print('# rocprofv3 summary %s (bench.py GHND Faster R-CNN b3ch, batch 16, 1 x MI355X)\n' % tag)
stats = glob.glob(os.path.join(out, 'trace', '*kernel_stats.csv'))
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    print('## kernel-trace --stats (2 timed + 1 warm-up steps)\n')
    print('| kernel | calls | total ms | avg us | % |')
    print('|---|---|---|---|---|')
    for r in rows[:24]:
        print('| %s | %s | %.2f | %.1f | %s |' % (short(r['Name']), r['Calls'], float(r['TotalDurationNs']) / 1e6,
                                                   float(r['AverageNs']) / 1e3, r['Percentage']))
for label, sub, col in (('FETCH_SIZE', 'pmc_fetch', 'FETCH_SIZE'), ('WRITE_SIZE', 'pmc_write', 'WRITE_SIZE')):
    files = glob.glob(os.path.join(out, sub, '*counter_collection.csv'))
    if not files:
        continue
    agg = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(files[0])):
        if r.get('Counter_Name') == col:
            a = agg[short(r['Kernel_Name'])]
            a[0] += float(r['Counter_Value'])
            a[1] += 1
    print('\n## %s per kernel (one warm-up + one timed step; counter unit KiB; on gfx950 FETCH_SIZE counts half '
          'the bytes of wide coalesced reads -> double it, MI355X_MICROARCH.md HBM section)\n' % label)
    print('| kernel | dispatches | sum KiB | avg MiB / dispatch |')
    print('|---|---|---|---|')
    for k, (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:12]:
        print('| %s | %d | %.0f | %.2f |' % (k, n, v, v / n / 1024.0))
for f in ('bench_trace.json',):
    p = os.path.join(out, f)
    if os.path.exists(p) and os.path.getsize(p):
        try:
            d = json.loads(open(p).read().strip().splitlines()[-1])
            print('\n## bench line under the profiler\n\n```\n%s\n```' % json.dumps({k: d[k] for k in ('value', 'ms_per_step', 'roofline', 'kernels')}))
        except Exception as e:      # noqa
            print('\n(bench json unreadable: %s)' % e)
