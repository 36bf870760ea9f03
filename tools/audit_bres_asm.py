#!/usr/bin/env python
"""Audit of the inline-asm register rings of bres2_kernel (csrc/conv_bres.hip) and bstream_kernel
(csrc/conv_bstream.hip) in hipcc's assembly output.

hipcc does not know that a `global_load_dwordx4` inside an asm statement writes its destination LATER: between that
statement and the `s_waitcnt vmcnt(N)` statement that names the same registers it is free to copy / spill / reuse them.
This script walks every such kernel in the .s file and fails if any instruction reads or writes a ring register
between its asm load and the asm wait that releases it (a second asm load into a still-pending register also fails).

usage: python tools/audit_bres_asm.py [file.s ...]      (without a file: compiles the two sources to assembly first)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r'\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b')


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def audit(path):
    kernels, cur, name = {}, None, None
    for line in open(path):
        m = re.match(r'^(_ZN\S*(?:bres2|bstream)_kernel\S*):', line)
        if m:
            name, cur = m.group(1), []
            kernels[name] = cur
        elif cur is not None:
            cur.append(line.rstrip('\n'))
            if line.startswith('.Lfunc_end'):     # (a kernel may hold several s_endpgm: early exits)
                cur = None
    problems, stats = [], {}
    for name, lines in kernels.items():
        pending = {}            # register -> line number of the asm load that targets it
        order = []              # asm loads still in flight, oldest first: (line, destination registers)
        in_asm, loads, waits = False, 0, 0

        def release(keep):
            """memory operations retire in issue order: after `vmcnt(keep)` all but the `keep` youngest have landed
            (the compiler's own loads / stores in between only make the real state more complete than this model)"""
            while len(order) > keep:
                _, dst = order.pop(0)
                for r in dst:
                    pending.pop(r, None)

        labels = {l.split(':')[0].strip(): i for i, l in enumerate(lines) if re.match(r'^\.LBB\S+:', l)}
        replayed = set()

        def scan(lo, hi, top):
            nonlocal in_asm, loads, waits
            for idx in range(lo, hi):
                raw, no = lines[idx], idx + 1
                text = raw.split(';')[0].strip() if not raw.strip().startswith(';') else ''
                if '#ASMSTART' in raw:
                    in_asm = True
                    continue
                if '#ASMEND' in raw:
                    in_asm = False
                    continue
                if not text or text.endswith(':') or text.startswith('.'):
                    continue
                m = re.match(r's_waitcnt.*vmcnt\((\d+)\)', text)
                if m:                                   # asm or compiler wait: same hardware counter
                    waits += in_asm and top
                    release(int(m.group(1)))
                    continue
                if in_asm and text.startswith('global_load_dwordx4'):
                    dst = regs_of(text.split(',')[0])
                    clash = dst & set(pending)
                    if clash:
                        problems.append('%s:%d asm load into still-pending %s' % (name[-40:], no, sorted(clash)[:4]))
                    srcs = regs_of(','.join(text.split(',')[1:]))
                    if srcs & set(pending):
                        problems.append('%s:%d asm load address uses a pending register' % (name[-40:], no))
                    for r in dst:
                        pending[r] = no
                    order.append((no, dst))
                    loads += top
                    continue
                if in_asm:
                    continue
                touched = regs_of(text) & set(pending)
                if touched:     # a compiler instruction reads / writes a register whose asm load may still be in flight
                    problems.append('%s:%d `%s` touches %s (asm load at line %d)'
                                    % (name[-40:], no, text[:60], sorted(touched)[:4], pending[sorted(touched)[0]]))
                b = re.match(r's_cbranch_\w+\s+(\.LBB\S+)|s_branch\s+(\.LBB\S+)', text)
                if b and top:
                    tgt = labels.get(b.group(1) or b.group(2))
                    if tgt is not None and tgt < idx and (tgt, idx) not in replayed:
                        # loop back edge: walk the body once more with what is in flight at the bottom of the loop
                        replayed.add((tgt, idx))
                        saved = (dict(pending), list(order))
                        scan(tgt, idx, False)
                        pending.clear(); pending.update(saved[0])
                        order[:] = saved[1]

        scan(0, len(lines), True)
        stats[name] = (loads, waits)
    return kernels, problems, stats


def main():
    paths = sys.argv[1:]
    if not paths:
        tmp = tempfile.mkdtemp()
        for stem in ('conv_bres', 'conv_bstream'):
            path = os.path.join(tmp, stem + '.s')
            src = os.path.join(ROOT, 'hnd_ghnd_object_detectors_amd', 'csrc', stem + '.hip')
            subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950',
                                   '-I' + os.path.join(ROOT, 'include'), '-ffp-contract=fast', '-S',
                                   '--cuda-device-only', src, '-o', path], stderr=subprocess.DEVNULL)
            paths.append(path)
    rc = 0
    for path in paths:
        kernels, problems, stats = audit(path)
        for k, (l, w) in stats.items():
            print('%s: %d asm loads, %d asm waits' % (k, l, w))
        if not kernels:
            print('%s: no ring kernel found' % path)
            rc = 1
        for p in problems[:40]:
            print('PROBLEM', p)
        print('%s: %d problem(s)' % (os.path.basename(path), len(problems)))
        rc = rc or (1 if problems else 0)
    return rc


if __name__ == '__main__':
    sys.exit(main())
