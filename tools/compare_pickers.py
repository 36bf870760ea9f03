#!/usr/bin/env python3
"""Which launches would another kernel choice serve better?  Reads per-launch tables of the step (`bench.py --detail`)
made under different global picker settings and lists, per launch tag, the settings that beat the default.

usage: python3 tools/compare_pickers.py default.txt NAME=other.txt [NAME=other2.txt ...] [--min_ms 0.01] [--min_rel 0.03]
"""
import collections
import sys


def load(path):
    per = collections.OrderedDict()
    for line in open(path):
        p = line.split()
        if len(p) < 6 or p[0] == 'launch':
            continue
        try:
            tag, kern, n, ms = p[0], p[1], int(p[2]), float(p[3])
        except ValueError:
            continue
        e = per.setdefault(tag, [0.0, []])
        e[0] += ms
        e[1].append('%s x%d' % (kern, n))
    return per


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    opts = dict(a[2:].split('=') for a in sys.argv[1:] if a.startswith('--') and '=' in a)
    min_ms, min_rel = float(opts.get('min_ms', 0.01)), float(opts.get('min_rel', 0.03))
    base = load(args[0])
    total = collections.Counter()
    for spec in args[1:]:
        name, path = spec.split('=', 1)
        other = load(path)
        print('== %s (total of the tagged launches: default %.2f ms, %s %.2f ms)'
              % (name, sum(v[0] for v in base.values()), name, sum(v[0] for v in other.values())))
        rows = []
        for tag, (ms, kerns) in base.items():
            if tag in other and other[tag][1] != kerns:
                d = ms - other[tag][0]
                rows.append((d, tag, ms, kerns, other[tag][0], other[tag][1]))
        for d, tag, ms, kerns, oms, okerns in sorted(rows, reverse=True):
            mark = '+' if (d > min_ms and d > min_rel * ms) else (' ' if abs(d) <= min_ms else '-')
            print(' %s %-30s %7.3f ms [%s] -> %7.3f ms [%s]  (%+.3f)' % (mark, tag, ms, ', '.join(kerns), oms,
                                                                       ', '.join(okerns), -d))
            if mark == '+':
                total[name] += d
    print('gains available (sum of "+" rows):', dict((k, round(v, 3)) for k, v in total.items()))


if __name__ == '__main__':
    main()
