#!/usr/bin/env python
"""Condense `rocprofv3 --pmc SQ_*` output (counter_collection.csv) into matrix-pipe utilisation per kernel.

    python tools/sq_counters.py <dir with *counter_collection.csv> [title]

Units (MI355X_MICROARCH.md, rocprofv3 PMC slots): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count QUAD-cycles summed
over the waves of a dispatch, SQ_VALU_MFMA_BUSY_CYCLES counts cycles.  With W resident waves per SIMD the SIMD-cycles of a
dispatch are 4 * WAVE_CYCLES / W, so
    matrix-pipe utilisation = MFMA_BUSY / (4 * WAVE_CYCLES / W)
W comes from the kernel: the one-wave-per-SIMD persistent GEMMs (bres2, bstream: 4 waves per workgroup, one workgroup per
CU) have W = 1, the 8-wave B-resident kernel W = 2, the tiled kernel 3 or 4 (blocks per CU).  WAIT_ANY = wave parked at
s_waitcnt / barrier; WAIT_INST_ANY = issue stall (mfma read-after-write, pipe busy)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def waves_per_simd(name):
    if 'bres2_kernel' in name or 'bstream_kernel' in name or 'wgrad_ring' in name or 'bx3_kernel' in name or 'bxs_kernel' in name:
        return 1
    if 'bres_kernel' in name:
        return 2
    m = re.search(r'igemm_kernel<(\d+), (\d+)', name)
    if m:
        return 3 if (m.group(1), m.group(2)) == ('128', '128') else 4
    if 'wgrad_kernel' in name:
        return 4
    return None


def short(name):
    name = name.replace('void ', '').replace('(anonymous namespace)::', '')
    return name.split('(')[0][:64]


def main():
    out = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else out
    files = sorted(glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True))
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in files:
        for r in csv.DictReader(open(f)):
            a = agg[r['Kernel_Name']][r['Counter_Name']]
            a[0] += float(r['Counter_Value'])
            a[1] += 1
    print('# %s' % title)
    print('# rocprofv3 --kernel-trace --pmc SQ_* (counters only; %d file(s)); mean per dispatch' % len(files))
    print('# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* = quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES = cycles')
    print('# matrix-pipe utilisation = MFMA_BUSY / (4 * WAVE_CYCLES / W), W = resident waves per SIMD of that kernel')
    for kname, ctrs in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', [0, 1])[0]):
        if 'SQ_WAVE_CYCLES' not in ctrs or 'SQ_VALU_MFMA_BUSY_CYCLES' not in ctrs:
            continue
        wc = ctrs['SQ_WAVE_CYCLES'][0] / ctrs['SQ_WAVE_CYCLES'][1]
        if ctrs['SQ_VALU_MFMA_BUSY_CYCLES'][0] == 0:
            continue
        print('%s   (%d dispatches)' % (short(kname), ctrs['SQ_WAVE_CYCLES'][1]))
        for c in sorted(ctrs):
            v = ctrs[c][0] / ctrs[c][1]
            print('   %-28s mean %.4g  (%.1f %% of WAVE_CYCLES)' % (c, v, 100.0 * v / wc))
        w = waves_per_simd(kname)
        mf = ctrs['SQ_VALU_MFMA_BUSY_CYCLES'][0] / ctrs['SQ_VALU_MFMA_BUSY_CYCLES'][1]
        if w:
            print('   matrix-pipe utilisation = MFMA_BUSY / (4 * WAVE_CYCLES / %d waves per SIMD) = %.3f' % (w, mf / (4.0 * wc / w)))
        if 'SQ_WAIT_ANY' in ctrs:
            print('   wave parked at s_waitcnt / barrier: %.1f %% of its life; issue-stalled: %.1f %%' % (
                100.0 * ctrs['SQ_WAIT_ANY'][0] / ctrs['SQ_WAIT_ANY'][1] / wc,
                100.0 * ctrs.get('SQ_WAIT_INST_ANY', [0, 1])[0] / ctrs.get('SQ_WAIT_INST_ANY', [0, 1])[1] / wc))


if __name__ == '__main__':
    main()
