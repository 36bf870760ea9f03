set -x
mkdir -p gpurun_out/r06b
for cfg in "HND_BX3_HEAD_FWD=0" "HND_BX3_HEAD_FWD=0 HND_BX3_STUDENT_FWD=0"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  env $cfg python -m pytest tests/test_model_gpu.py -q -p no:cacheprovider -k "full_size_step or dense_parity or batch16" > gpurun_out/r06b/tests_$tag.txt 2>&1
  echo "rc=$?" >> gpurun_out/r06b/tests_$tag.txt
  env $cfg python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_runner --no_native_leg > gpurun_out/r06b/bench_$tag.json 2> gpurun_out/r06b/bench_$tag.err
done
python -m pytest tests/test_bx3_gpu.py -q -p no:cacheprovider > gpurun_out/r06b/tests_bx3.txt 2>&1
tail -5 gpurun_out/r06b/tests_*.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06b/bench_*.json')):
    try:
        j=json.loads([l for l in open(f) if l.startswith('{"metric"')][-1]); print(f, j['value'], j['ms_per_step'], j['value_resident'])
    except Exception as e: print(f, 'ERR', e)
PY
