#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of every Winograd transform launch of tools/bench_wino_transforms.py (separate PMC passes), beside
# the bytes the transform must move: does the input transform's 8x8-patch halo (stride 6: every pixel is read 1.78 times)
# come out of L2 or out of HBM?   usage (GPU box): bash tools/wino_transform_traffic.sh gpurun_out/wino_traffic
set -e
OUT=${1:-gpurun_out/wino_traffic}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 tools/bench_wino_transforms.py --reps 20 > $OUT/timing.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o w -- python3 tools/bench_wino_transforms.py --reps 2 > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 tools/bench_wino_transforms.py --reps 2 > $OUT/write.log 2>&1
python3 - $OUT <<'P'
import csv, sys, os, collections
out = sys.argv[1]
def load(d, name):
    rows = collections.OrderedDict()
    for root, _, files in os.walk(os.path.join(out, d)):
        for f in files:
            if f.endswith('counter_collection.csv'):
                for r in csv.DictReader(open(os.path.join(root, f))):
                    if r['Counter_Name'] == name and 'wino' in r['Kernel_Name']:
                        rows[int(r['Dispatch_Id'])] = (r['Kernel_Name'].split('(')[0][-40:], int(r['Grid_Size']), float(r['Counter_Value']))
    return rows
fe, wr = load('fetch', 'FETCH_SIZE'), load('write', 'WRITE_SIZE')
print('%-44s %10s %12s %12s' % ('kernel (dispatch order; 3 warm-up + 2 timed launches per transform)', 'grid', 'fetch MB', 'write MB'))
for (k, (name, grid, f)), (k2, (n2, g2, w)) in zip(fe.items(), wr.items()):
    print('%-44s %10d %12.1f %12.1f' % (name, grid, f * 2 * 1024 / 1e6, w * 1024 / 1e6))      # FETCH_SIZE: KiB of 64-B halves on gfx950 -> x2
P
