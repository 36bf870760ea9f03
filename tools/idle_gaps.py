#!/usr/bin/env python3
"""Where does the step go when no kernel runs?  Reads a rocprofv3 --kernel-trace CSV of `bench.py` (default streams: the
teacher / student / FPN chains overlap) and reports, over the window of the last `--steps` steps:

  wall time, union of the kernel intervals (GPU busy), idle = wall - union, overlap = sum of durations - union,
  a histogram of the idle gaps and the kernels that most often FOLLOW a long gap.

usage (on the GPU box):
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps -o bench -- python3 bench.py --steps 6 --warmup 2 --no_cpu_baseline
  python3 tools/idle_gaps.py gpurun_out/gaps/*/bench_kernel_trace.csv --steps 4 --marker adam
"""
import argparse
import collections
import csv
import re


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name.split('(')[0][:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--steps', type=int, default=4, help='steps in the window')
    ap.add_argument('--skip_last', type=int, default=2,
                    help='steps at the end to leave out (bench.py ends with its event-timed, serialised roofline steps)')
    ap.add_argument('--marker', default='adam', help='substring of the kernel that ends a step')
    ap.add_argument('--long_us', type=float, default=5.0)
    a = ap.parse_args()
    rows = []
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if a.marker in r[2].lower()]
    assert len(ends) > a.steps + a.skip_last, 'not enough step markers (%d)' % len(ends)
    ends = ends[:len(ends) - a.skip_last]
    lo, hi = ends[-a.steps - 1] + 1, ends[-1] + 1
    win = rows[lo:hi]
    t0, t1 = win[0][0], max(r[1] for r in win)
    wall = (t1 - t0) / 1e6
    total = sum(r[1] - r[0] for r in win) / 1e6
    union, gaps = 0, []
    cur_s, cur_e = win[0][0], win[0][1]
    for s, e, n in win[1:]:
        if s > cur_e:
            union += cur_e - cur_s
            gaps.append(((s - cur_e) / 1e3, n))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    union /= 1e6
    k = a.steps
    print('window: %d steps, %d kernel launches (%.0f per step)' % (k, len(win), len(win) / k))
    print('per step: wall %.3f ms | GPU busy (union) %.3f | idle %.3f | sum of kernel durations %.3f | overlapped %.3f'
          % (wall / k, union / k, (wall - union) / k, total / k, (total - union) / k))
    edges = [0, 1, 2, 3, 5, 10, 20, 50, 100, 1e9]
    hist = collections.OrderedDict()
    for lo_, hi_ in zip(edges[:-1], edges[1:]):
        g = [x for x, _ in gaps if lo_ <= x < hi_]
        hist['%g-%g us' % (lo_, hi_)] = (len(g) / k, sum(g) / 1e3 / k)
    print('idle gaps per step (count, ms):')
    for name, (c, ms) in hist.items():
        print('  %-12s %7.1f  %7.3f' % (name, c, ms))
    after = collections.Counter()
    after_ms = collections.Counter()
    for g, n in gaps:
        if g >= a.long_us:
            after[short(n)] += 1
            after_ms[short(n)] += g / 1e3
    print('kernels that start after a gap >= %g us (count per step, ms per step):' % a.long_us)
    for n, ms in after_ms.most_common(15):
        print('  %-60s %6.1f  %7.3f' % (n, after[n] / k, ms / k))


if __name__ == '__main__':
    main()
