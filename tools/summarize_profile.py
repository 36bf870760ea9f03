#!/usr/bin/env python
"""Condense rocprofv3 output of tools/profile_round.sh into a small markdown summary (profiles/rNN_summary.md)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]


import re

FULLNAMES = True


def short(name):
    m = re.search(r'igemm_kernel<(\d+), (\d+), (\d+), (true|false), (true|false)>', name)
    if m:
        base = ('igemm_c4_%sx%s' if m.group(4) == 'true' else 'igemm_%sx%s') % (m.group(1), m.group(2))
        return base + ('[bn-prologue]' if m.group(5) == 'true' and FULLNAMES else '')
    m = re.search(r'bres(2?)_kernel<(\d+), (\d+), (true|false)(?:, (true|false))?(?:, \d+)?>', name)
    if m:           # B-resident persistent GEMM (conv_bres.hip): wave columns -> width of the resident weight slice
        tag = ('[prologue]' if m.group(4) == 'true' else '') + ('[res]' if m.group(5) == 'true' else '')
        return 'bres%s_%d' % (m.group(1), 64 * int(m.group(2))) + (tag if FULLNAMES else '')
    m = re.search(r'bstream_kernel<(\d+), (true|false), (true|false)>', name)
    if m:           # B-streamed persistent GEMM (conv_bstream.hip)
        return 'bstream_%d' % (64 * int(m.group(1))) + (('[prologue]' if m.group(2) == 'true' else '') +
                                                      ('[taps]' if m.group(3) == 'true' else '') if FULLNAMES else '')
    m = re.search(r'bx3_kernel<(\d+), (true|false), (true|false)(?:, (true|false))?>', name)
    if m:           # fp32 emulated on the bf16 pipe, B resident (conv_bx3.hip): k steps of 32, residual, mask out, mask in
        return 'bx3_64' + ('[K%d%s%s%s]' % (32 * int(m.group(1)), ', res' if m.group(2) == 'true' else '',
                                            ', bits out' if m.group(3) == 'true' else '', ', bits in' if m.group(4) == 'true' else '')
                           if FULLNAMES else '')
    m = re.search(r'bxs_kernel<(\d+), (true|false), (true|false)>', name)
    if m:           # ... B streamed (conv_bxs.hip)
        return 'bxs_%d' % (64 * int(m.group(1))) + (('[prologue]' if m.group(2) == 'true' else '') +
                                                  ('[taps]' if m.group(3) == 'true' else '') if FULLNAMES else '')
    m = re.search(r'wgrad_ring_kernel<(\d+), (\d+), (true|false), (\d+)>', name)
    if m:           # ring weight gradient (conv_wgrad_ring.hip): wave tile 64 AH x 64 BH, taps, waves per SIMD
        return 'wgrad_ring' + ('[%dx%d%s]' % (64 * int(m.group(1)), 64 * int(m.group(2)), ', taps' if m.group(3) == 'true' else '')
                               if FULLNAMES else '')
    m = re.search(r'wgrad_kernel<(\d+), (\d+)>', name)
    if m:
        return 'wgrad_m%s' % m.group(1)
    for key, lab in (('igemm_kernel<128, false>', 'igemm_n128'), ('igemm_kernel<64, false>', 'igemm_n64'),
                     ('igemm_kernel<64, true>', 'igemm_c4_n64'), ('wgrad_kernel<128>', 'wgrad_m128'),
                     ('wgrad_kernel<64>', 'wgrad_m64'), ('igemm_kernelILi128ELb0', 'igemm_n128'),
                     ('igemm_kernelILi64ELb0', 'igemm_n64'), ('igemm_kernelILi64ELb1', 'igemm_c4_n64'),
                     ('wgrad_kernelILi128', 'wgrad_m128'), ('wgrad_kernelILi64', 'wgrad_m64')):
        if key in name:
            return lab
    if 'stem7_kernel' in name:
        return 'stem7_lds'
    for key in ('wino6_input', 'wino6_output', 'wino6_weights', 'wino4_input', 'wino4_output', 'wino4_weights', 'wino2_input', 'wino2_output', 'wino2_weights',
                'wino_input', 'wino_output', 'wino_weights'):
        if key + '_kernel' in name:
            return key
    name = name.replace('void ', '').replace('(anonymous namespace)::', '').replace('_ZN12_GLOBAL__N_1', '')
    return name.split('(')[0][:48]


def steps_label(json_name, default):
    """what the profiled bench.py process executed: its W warm-up (the first of them is the gated step) + K timed steps,
    then 3 enqueue-probe steps and 1 single-stream event-profiled step -- the call counts below cover ALL of them"""
    try:
        d = json.loads(open(os.path.join(out, json_name)).read().strip().splitlines()[-1])
        w, k = int(d['warmup']), int(d['steps'])
        return '%d steps in the process: %d warm-up + %d timed + 3 enqueue-probe + 1 event-profiled' % (w + k + 4, w, k)
    except Exception:      # noqa
        return default


print('# rocprofv3 summary %s (bench.py GHND Faster R-CNN b3ch, batch 16, 1 x MI355X)\n' % tag)
def stats_table(sub, json_name, title):
    """one kernel-stats table + the sum of ALL kernel durations per step next to bench.py's own per-step kernel times"""
    stats = glob.glob(os.path.join(out, sub, '*kernel_stats.csv'))
    if not stats:
        return
    rows = list(csv.DictReader(open(stats[0])))
    print('## kernel-trace --stats, %s (%s)\n' % (title, steps_label(json_name, 'steps unknown')))
    try:
        d = json.loads(open(os.path.join(out, json_name)).read().strip().splitlines()[-1])
        nsteps = int(d['warmup']) + int(d['steps']) + 4
        total = sum(float(r['TotalDurationNs']) for r in rows) / 1e6
        print('sum of all kernel durations %.1f ms = %.2f ms per step over %d steps; bench.py in the same process: wall %.2f '
              'ms/step, per-launch HIP events of its single-stream step: MFMA kernels %.2f + HBM-bound kernels %.2f = %.2f ms '
              '(the trace also holds the small launches bench.py does not time: packs, folds, finalizes)\n'
              % (total, total / nsteps, nsteps, d['ms_per_step'], d['conv_kernel_ms_per_step'], d['hbm_kernel_ms_per_step'],
                 d['conv_kernel_ms_per_step'] + d['hbm_kernel_ms_per_step']))
    except Exception as e:      # noqa
        print('(bench json unreadable: %s)\n' % e)
    print('| kernel | calls | total ms | avg us | % |')
    print('|---|---|---|---|---|')
    for r in rows[:24]:
        print('| %s | %s | %.2f | %.1f | %s |' % (short(r['Name']), r['Calls'], float(r['TotalDurationNs']) / 1e6,
                                                   float(r['AverageNs']) / 1e3, r['Percentage']))
    print()


stats_table('trace', 'bench_trace.json', 'SINGLE stream (HND_TEACHER_STREAM=0 HND_DEFER_FPN=0 HND_WGRAD_STREAM=0): kernels run '
            'alone -- per-kernel durations and fractions are read from THIS table')
stats_table('trace_ms', 'bench_trace_ms.json', 'the step as it runs (teacher, pyramid and weight-gradient side streams ON): '
            'co-running kernels are time-sliced, their durations are inflated -- do not compute per-kernel fractions from it')
FULLNAMES = False      # traffic is keyed by the bench variant names
traffic = {}
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
    print('\n## %s per kernel (%s; counter unit KiB; on gfx950 FETCH_SIZE counts half '
          'the bytes of wide coalesced reads -> double it, MI355X_MICROARCH.md HBM section)\n'
          % (label, steps_label('bench_fetch.json' if label == 'FETCH_SIZE' else 'bench_write.json',
                                '6 steps in the process: 1 warm-up + 1 timed + 3 enqueue-probe + 1 event-profiled')))
    traffic.setdefault(label, agg)
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

if 'FETCH_SIZE' in traffic and 'WRITE_SIZE' in traffic:
    tj = {'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `bench.py --steps 1 '
                    '--warmup 1`, summed over all 6 steps that process runs (1 warm-up + 1 timed + 3 enqueue-probe + 1 '
                    'event-profiled) and divided by the dispatch count; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts '
                    '64 B per 128-B request)', 'unit': 'bytes per kernel dispatch (a K > 256 launch of the B-resident emulation kernel is several dispatches: one per pass over k)',
          'steps_in_process': 6, 'kernels': {}}
    for k, (fv, fn) in traffic['FETCH_SIZE'].items():
        if k in traffic['WRITE_SIZE'] and any(t in k for t in ('igemm', 'wgrad', 'bres', 'bx3', 'bxs', 'bstream', 'stem7')):
            wv, wn = traffic['WRITE_SIZE'][k]
            tj['kernels'][k] = {'dispatches': fn, 'fetch_kib_raw': fv, 'write_kib': wv,
                                'traffic_bytes_per_launch': int((2 * fv + wv) * 1024 / fn)}
    try:        # the configuration these counters belong to (bench.py only quotes them for a matching run)
        tj['config'] = json.loads(open(os.path.join(out, 'bench_trace.json')).read().strip().splitlines()[-1])['run_cfg']
    except Exception:      # noqa
        tj['config'] = None
    json.dump(tj, open(os.path.join(out, 'traffic.json'), 'w'), indent=1)

fstats = glob.glob(os.path.join(out, 'filter', '*kernel_stats.csv'))
if fstats:
    rows = list(csv.DictReader(open(fstats[0])))
    print('\n## neural filter (SURVEY 8f-f2): kernel-trace --stats of tools/bench_filter.py --batch 16 (10 + 5 steps)\n')
    print('| kernel | calls | total ms | avg us | % |')
    print('|---|---|---|---|---|')
    for r in rows[:16]:
        print('| %s | %s | %.2f | %.1f | %s |' % (short(r['Name']), r['Calls'], float(r['TotalDurationNs']) / 1e6,
                                                   float(r['AverageNs']) / 1e3, r['Percentage']))
    p = os.path.join(out, 'bench_filter.json')
    if os.path.exists(p) and os.path.getsize(p):
        print('\n```\n%s\n```' % open(p).read().strip().splitlines()[-1])
