# which of the head's forward convs may run emulated without turning a marginal parity test red (r06 experiment)
mkdir -p gpurun_out/r06e
for hf in "1" "0,1,5" "0,1,2,5,6" "0,2"; do
  tag=$(echo "$hf" | tr ',' '_')
  HND_BX3_HEAD_FWD=$hf python -m pytest tests/test_model_gpu.py -q -p no:cacheprovider -k "full_size_step or dense_parity or batch16" > gpurun_out/r06e/tests_hf_$tag.txt 2>&1
  echo "rc=$?" >> gpurun_out/r06e/tests_hf_$tag.txt
  HND_BX3_HEAD_FWD=$hf python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_runner --no_native_leg > gpurun_out/r06e/bench_hf_$tag.json 2> gpurun_out/r06e/bench_hf_$tag.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06e/bench_*.json')):
    try:
        j=json.loads([l for l in open(f) if l.startswith('{"metric"')][-1]); print(f, j['value'], j['ms_per_step'], j['value_resident'])
    except Exception as e: print(f, 'ERR', e)
for f in sorted(glob.glob('gpurun_out/r06e/tests_*.txt')):
    print(f, [l.strip() for l in open(f) if 'passed' in l or 'failed' in l or l.startswith('FAILED')])
PY
