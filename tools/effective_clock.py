#!/usr/bin/env python
"""Effective shader clock per kernel family from a rocprofv3 `--kernel-trace --pmc GRBM_GUI_ACTIVE` run of bench.py
(/opt/skills/guides/MI355X_MICROARCH.md, DVFS give-back: clock ~= GRBM_GUI_ACTIVE / 8 / kernel wall time -- rocprofv3 reports
the sum over the 8 XCDs; the quotient reads high on dispatches shorter than ~0.3 ms, so only dispatches of >= 0.3 ms count).

    python tools/effective_clock.py gpurun_out/prof_r05/pmc_clock [label]

Prints a markdown table: family, dispatches counted, mean duration, effective GHz, and what a fraction of the 2.4 GHz
nominal peak becomes against the peak at that clock."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def family(name):
    for key, lab in (('bx3_kernel', 'bx3_64 (bf16x3 emulation, B resident)'), ('bxs_kernel', 'bxs (bf16x3 emulation, B streamed)'), ('bres2_kernel', 'bres2'), ('bres_kernel', 'bres'),
                     ('bstream_kernel', 'bstream'), ('wgrad_ring_kernel', 'wgrad_ring'), ('igemm_kernel', 'igemm (tiled)'),
                     ('wgrad_kernel', 'wgrad (LDS-staged)'), ('stem7_wgrad', 'stem7_wgrad'), ('stem7_kernel', 'stem7_lds'),
                     ('wino6_input', 'wino6_input'), ('wino6_output', 'wino6_output'), ('wino26_', 'wino26_*'),
                     ('mse_kernel', 'mse'), ('maxpool', 'maxpool'), ('bn_bwd', 'bn_bwd_*'), ('affine_relu', 'affine_relu')):
        if key in name:
            return lab
    return None


def main():
    d = sys.argv[1]
    label = sys.argv[2] if len(sys.argv) > 2 else ''
    cc = glob.glob(os.path.join(d, '*counter_collection.csv'))
    kt = glob.glob(os.path.join(d, '*kernel_trace.csv'))
    if not cc:
        print('no counter_collection.csv under', d)
        return 1
    dur = {}
    if kt:
        for r in csv.DictReader(open(kt[0])):
            dur[r.get('Dispatch_Id')] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    agg = defaultdict(lambda: [0.0, 0.0, 0])
    for r in csv.DictReader(open(cc[0])):
        if r.get('Counter_Name') != 'GRBM_GUI_ACTIVE':
            continue
        if r.get('Start_Timestamp') and r.get('End_Timestamp'):
            ns = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
        else:
            ns = dur.get(r.get('Dispatch_Id'))
        fam = family(r['Kernel_Name'])
        if fam is None or not ns or ns < 0.3e6:
            continue
        a = agg[fam]
        a[0] += float(r['Counter_Value'])
        a[1] += ns
        a[2] += 1
    print('## effective clock per kernel family%s (GRBM_GUI_ACTIVE / 8 / wall, dispatches >= 0.3 ms)\n' % (' -- ' + label if label else ''))
    print('| family | dispatches | mean ms | effective GHz | a fraction f of the 2.4 GHz peak is f x this of the peak AT that clock |')
    print('|---|---|---|---|---|')
    for fam, (c, ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        ghz = c / 8.0 / ns
        print('| %s | %d | %.3f | %.2f | x%.2f |' % (fam, n, ns / n / 1e6, ghz, 2.4 / ghz))
    return 0


if __name__ == '__main__':
    sys.exit(main())
