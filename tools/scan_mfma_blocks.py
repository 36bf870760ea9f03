#!/usr/bin/env python
"""Do the MFMA kernels issue their matrix instructions in long straight runs?

Compiles every csrc/*.hip that holds MFMAs to assembly and prints, per kernel, how its v_mfma instructions are spread over
basic blocks.  A wave-uniform run-time test inside an unrolled MFMA loop (`if (four) mfma(...)`, `if (nown == 3) ...`) does
not cost a divergent branch, but it does end the basic block: hipcc emits `s_cbranch` + waits and the matrix pipe restarts --
the stem kernels ran at 0.59-0.66 of peak with one such branch behind every 2-12 MFMAs until round 5 made the tests
compile-time (NOTEBOOK section 11).  Many blocks of <= 16 MFMAs in a hot loop are the signature.

    python tools/scan_mfma_blocks.py            # all kernels with >= 32 MFMAs
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def scan(path):
    lines = open(path).read().split('\n')
    out = []
    for start in (i for i, l in enumerate(lines) if re.match(r'^_Z\S+:', l)):
        name = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', lines[start].split(':')[0])
        ends = [i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end')]
        if not ends:
            continue
        blocks, cur = [], 0
        for l in lines[start:ends[0]]:
            t = l.split(';')[0].strip()
            if re.match(r'^\.LBB', t):
                if cur:
                    blocks.append(cur)
                cur = 0
            elif t.startswith('v_mfma'):
                cur += 1
        if cur:
            blocks.append(cur)
        if sum(blocks) >= 32:
            out.append((name, sum(blocks), len(blocks), sum(1 for b in blocks if b <= 16), max(blocks)))
    return out


def main():
    tmp = tempfile.mkdtemp()
    rc = 0
    for src in sorted(glob.glob(os.path.join(ROOT, 'hnd_ghnd_object_detectors_amd', 'csrc', '*.hip'))):
        if 'mfma' not in open(src).read():
            continue
        dst = os.path.join(tmp, os.path.basename(src)[:-4] + '.s')
        subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950',
                               '-I' + os.path.join(ROOT, 'include'), '-ffp-contract=fast', '-S', '--cuda-device-only', src,
                               '-o', dst], stderr=subprocess.DEVNULL)
        for name, tot, nb, small, mx in scan(dst):
            flag = '  <-- short runs' if small > 16 else ''
            print('%-18s %-72s mfma %5d  blocks %3d  of <= 16: %3d  longest %4d%s'
                  % (os.path.basename(src), name[:72], tot, nb, small, mx, flag))
            rc = rc or (1 if small > 16 else 0)
    return rc


if __name__ == '__main__':
    sys.exit(main())
