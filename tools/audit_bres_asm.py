#!/usr/bin/env python
"""Audit of the inline-asm register rings of bres2_kernel (csrc/conv_bres.hip), bstream_kernel
(csrc/conv_bstream.hip), wgrad_ring_kernel (csrc/conv_wgrad_ring.hip), bx3_kernel (csrc/conv_bx3.hip, the bf16x3
emulation) and bxs_kernel (csrc/conv_bxs.hip, its B-streamed build: the LDS-DMA pieces of the weight stages sit in the same
in-order queue) in hipcc's assembly output.

hipcc does not know that a `global_load_dwordx4` inside an asm statement writes its destination LATER: between that
statement and the `s_waitcnt vmcnt(N)` statement that names the same registers it is free to copy / spill / reuse them.
This script walks the control-flow graph of every such kernel in the .s file (every path, loop back edges included, each
basic block once per distinct set of loads in flight) and fails if any instruction reads or writes a ring register
between its asm load and the wait that releases it (a second asm load into a still-pending register also fails).
The three fp32 kernels' counts ignore their stores (a store in the window only makes a wait stronger) and so does the walk;
bx3_kernel's counts are EXACT -- they name the previous tile's stores (vector memory operations of one wave retire in issue
order through the one counter; LLVM's own wait insertion relies on the same on gfx9) -- so for it the walk puts every
compiler-emitted global store / load into the in-order queue as well.

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
        m = re.match(r'^(_ZN\S*(?:bres2|bstream|wgrad_ring|(?<!pack_)bx3|(?<!pack_)bxs)_kernel\S*):', line)
        if m:
            name, cur = m.group(1), []
            kernels[name] = cur
        elif cur is not None:
            cur.append(line.rstrip('\n'))
            if line.startswith('.Lfunc_end'):     # (a kernel may hold several s_endpgm: early exits)
                cur = None
    problems, stats = [], {}
    for name, lines in kernels.items():
        # ---- basic blocks: a label starts one, a branch / s_endpgm ends one
        label_at = {}
        for i, l in enumerate(lines):
            m = re.match(r'^(\.LBB\S+):', l)
            if m:
                label_at[m.group(1)] = i
        starts = sorted(set([0] + list(label_at.values())))
        loads, waits = 0, 0
        seen, found = set(), set()
        exact = 'bx3_kernel' in name            # stores / compiler loads take their place in the in-order queue

        def step_block(lo, state):
            """simulate lines from `lo` to the end of its block; returns [(next line index, state)]"""
            nonlocal loads, waits
            pending, order = dict(state[0]), list(state[1])
            in_asm = False
            idx = lo
            while idx < len(lines):
                raw, no = lines[idx], idx + 1
                if idx != lo and idx in starts_set:         # fell into the next block
                    return [(idx, (pending, order))]
                text = raw.split(';')[0].strip() if not raw.strip().startswith(';') else ''
                if '#ASMSTART' in raw:
                    in_asm = True
                elif '#ASMEND' in raw:
                    in_asm = False
                elif text and not text.endswith(':') and not text.startswith('.'):
                    m = re.match(r's_waitcnt.*vmcnt\((\d+)\)', text)
                    if m:                                   # asm or compiler wait: same hardware counter; memory
                        keep = int(m.group(1))              # operations retire in issue order
                        while len(order) > keep:
                            _, dst = order.pop(0)
                            for r in dst:
                                pending.pop(r, None)
                    elif in_asm and exact and re.match(r'global_(load_dword|load_ubyte|store_)\S*\s', text) and not text.startswith('global_load_dwordx4'):
                        # an asm store: a place in the queue; a throw-away 4-byte load or a mask-byte load: its register is
                        # pending as well
                        dst = regs_of(text.split(',')[0]) if text.startswith('global_load') else set()
                        for r in dst:
                            pending[r] = no
                        order.append((no, frozenset(dst)))
                    elif in_asm and text.startswith('global_load_lds_'):
                        # LDS-DMA (bxs_kernel's weight stages): no register, but a place in the in-order queue that the
                        # kernel's counts name
                        order.append((no, frozenset()))
                    elif in_asm and text.startswith('global_load_dwordx4'):
                        dst = regs_of(text.split(',')[0])
                        clash = dst & set(pending)
                        if clash:
                            found.add('%s:%d asm load into still-pending %s' % (name[-40:], no, sorted(clash)[:4]))
                        if regs_of(','.join(text.split(',')[1:])) & set(pending):
                            found.add('%s:%d asm load address uses a pending register' % (name[-40:], no))
                        for r in dst:
                            pending[r] = no
                        order.append((no, frozenset(dst)))
                    elif not in_asm:
                        if exact and re.match(r'global_(store|load)_', text):
                            order.append((no, frozenset()))
                        touched = regs_of(text) & set(pending)
                        if touched:     # a compiler instruction reads / writes a register whose asm load may be in flight
                            found.add('%s:%d `%s` touches %s (asm load at line %d)'
                                      % (name[-40:], no, text[:60], sorted(touched)[:4], pending[sorted(touched)[0]]))
                        if text.startswith('s_endpgm'):
                            return []
                        b = re.match(r's_branch\s+(\.LBB\S+)', text)
                        if b:
                            return [(label_at[b.group(1)], (pending, order))] if b.group(1) in label_at else []
                        # `s_cbranch_execnz L; s_branch M` is how hipcc closes the arm of a wave-uniform if / else: the
                        # fall-through would need EXEC == 0, and a wave that runs this code has live lanes (the kernels have no
                        # lane-divergent control flow: a wave is inside `if (cc < seg_hi)` with all of its lanes or not at all)
                        b = re.match(r's_cbranch_execnz\s+(\.LBB\S+)', text)
                        if b and b.group(1) in label_at:
                            return [(label_at[b.group(1)], (pending, order))]
                        b = re.match(r's_cbranch_\w+\s+(\.LBB\S+)', text)
                        if b and b.group(1) in label_at:
                            return [(label_at[b.group(1)], (dict(pending), list(order))), (idx + 1, (pending, order))]
                idx += 1
            return []

        starts_set = set(starts)
        for l in lines:                                     # counts for the report
            t = l.split(';')[0].strip()
            loads += t.startswith('global_load_dwordx4') and 1 or 0
        work = [(0, ({}, []))]
        while work:
            lo, state = work.pop()
            key = (lo, tuple(sorted(state[0].items())), tuple(state[1]))
            if key in seen:
                continue
            seen.add(key)
            if len(seen) > 20000:
                found.add('%s: state space of the walk exploded' % name[-40:])
                break
            work.extend(step_block(lo, state))
        problems.extend(sorted(found))
        waits = sum(1 for l in lines if re.match(r'\s*s_waitcnt vmcnt\(\d+\)\s*$', l))
        stats[name] = (loads, waits)
    return kernels, problems, stats


def main():
    paths = sys.argv[1:]
    if not paths:
        tmp = tempfile.mkdtemp()
        procs = []
        for stem in ('conv_bres', 'conv_bstream', 'conv_wgrad_ring', 'conv_bx3', 'conv_bxs'):      # (the five compile side by side)
            path = os.path.join(tmp, stem + '.s')
            src = os.path.join(ROOT, 'hnd_ghnd_object_detectors_amd', 'csrc', stem + '.hip')
            procs.append(subprocess.Popen(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950',
                                           '-I' + os.path.join(ROOT, 'include'), '-ffp-contract=fast', '-S',
                                           '--cuda-device-only', src, '-o', path], stderr=subprocess.DEVNULL))
            paths.append(path)
        for pr in procs:
            if pr.wait() != 0:
                raise subprocess.CalledProcessError(pr.returncode, pr.args)
    rc = 0
    for path in paths:
        kernels, problems, stats = audit(path)
        for k, (l, w) in stats.items():
            print('%s: %d 16-byte loads, %d bare vmcnt waits' % (k, l, w))
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
