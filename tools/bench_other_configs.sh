#!/bin/bash
# Run on the GPU box (via gpurun): the other BASELINE.json configurations on the current build, each with its per-term oracle
# gate where one is pinned (tests/golden/bench_first_loss.json) and value_native_fp32 beside it.
# usage: tools/bench_other_configs.sh r06   ->  gpurun_out/other_r06/r06_bench_other_configs.json
set -u
R=${1:-r06}
OUT=gpurun_out/other_$R
mkdir -p $OUT
run() {   # name, args...
  name=$1; shift
  python3 bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_runner --detail $OUT/detail_$name.txt "$@" > $OUT/$name.json 2> $OUT/$name.err
  echo "$name rc=$?"
}
run hnd_b16 --method hnd
run mask_b8 --model mask_rcnn --batch 8
run keypoint_b8 --model keypoint_rcnn --batch 8
run ghnd_b4 --batch 4
run ghnd_b16 
python3 - "$OUT" "$R" <<'PY'
import json, sys, os
out, r = sys.argv[1], sys.argv[2]
res = {}
for name in ('hnd_b16', 'mask_b8', 'keypoint_b8', 'ghnd_b4', 'ghnd_b16'):
    try:
        j = json.loads([l for l in open(os.path.join(out, name + '.json')) if l.startswith('{"metric"')][-1])
    except Exception as exc:
        res[name] = {'error': '%s: %s' % (type(exc).__name__, exc)}
        continue
    res[name] = {k: j.get(k) for k in ('value', 'unit', 'ms_per_step', 'value_resident', 'value_native_fp32', 'steps', 'warmup', 'dtype')}
    res[name]['workload'] = j['config']['workload']
    res[name]['loss_check_worst_rel_err'] = (j.get('loss_check') or {}).get('worst_rel_err')
    res[name]['emulation'] = {k: (j.get('emulation') or {}).get(k) for k in ('launches', 'ms', 'native_mfma_launches', 'native_mfma_ms')}
    res[name]['conv_kernel_ms_per_step'] = j.get('conv_kernel_ms_per_step')
    res[name]['hbm_kernel_ms_per_step'] = j.get('hbm_kernel_ms_per_step')
    res[name]['kernels'] = j.get('kernels')
if 'value' in res.get('ghnd_b4', {}) and 'value' in res.get('ghnd_b16', {}):
    res['batch4_over_batch16'] = round(res['ghnd_b4']['value'] / res['ghnd_b16']['value'], 4)
json.dump(res, open(os.path.join(out, r + '_bench_other_configs.json'), 'w'), indent=1)
for k, v in res.items():
    print(k, v if not isinstance(v, dict) else {a: v.get(a) for a in ('value', 'ms_per_step', 'value_native_fp32', 'loss_check_worst_rel_err', 'error')})
PY
