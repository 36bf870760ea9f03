#!/bin/bash
# Same-box A/B of the shared-trunk start layer and the picker thresholds that go with it (run via gpurun).
# usage: tools/sweep_trunk.sh OUTDIR
OUT=${1:-gpurun_out/sweep}
mkdir -p $OUT
run() {   # name, env assignments...
  name=$1; shift
  env "$@" python bench.py --steps 10 --warmup 3 --no_cpu_baseline --detail $OUT/detail_$name.txt > $OUT/bench_$name.json 2>> $OUT/bench.err
  python - "$OUT/bench_$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print('%-40s %8.2f img/s %8.3f ms  mfma %.2f ms  hbm %.2f ms' % (sys.argv[2], d['value'], d['ms_per_step'], d['conv_kernel_ms_per_step'], d['hbm_kernel_ms_per_step']))
except Exception as e:
    print('%-40s FAILED %s' % (sys.argv[2], e))
PY
}
run sep_a            HND_MERGE_TRUNK=0
run l3               HND_MERGE_FROM=layer3
run l3_k8            HND_MERGE_FROM=layer3 HND_BSTREAM_K1024_TILES=8
run l3_k8_w32        HND_MERGE_FROM=layer3 HND_BSTREAM_K1024_TILES=8 HND_BRES2_WINO_MIN=32
run l3_k4_w32        HND_MERGE_FROM=layer3 HND_BSTREAM_K1024_TILES=4 HND_BRES2_WINO_MIN=32
run l2_k8_w32        HND_MERGE_FROM=layer2 HND_BSTREAM_K1024_TILES=8 HND_BRES2_WINO_MIN=32
run l4_k4            HND_MERGE_FROM=layer4 HND_BSTREAM_K1024_TILES=4
run sep_k4           HND_MERGE_TRUNK=0 HND_BSTREAM_K1024_TILES=4
run sep_b            HND_MERGE_TRUNK=0
if [ "${2:-}" = "more" ]; then
run l3_nodefer       HND_MERGE_FROM=layer3 HND_BSTREAM_K1024_TILES=4 HND_BRES2_WINO_MIN=32 HND_DEFER_FPN=0
run l3_noteacher     HND_MERGE_FROM=layer3 HND_BSTREAM_K1024_TILES=4 HND_BRES2_WINO_MIN=32 HND_TEACHER_STREAM=0
run sep_single       HND_MERGE_TRUNK=0 HND_TEACHER_STREAM=0 HND_DEFER_FPN=0
run l2_single        HND_MERGE_FROM=layer2 HND_BSTREAM_K1024_TILES=4 HND_BRES2_WINO_MIN=32 HND_TEACHER_STREAM=0 HND_DEFER_FPN=0
fi
