#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + HBM traffic counters of the bench step.
# usage: tools/profile_round.sh r01
set -u
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$R
mkdir -p $OUT
# (1) the step as it runs: teacher / pyramid / weight-gradient side streams on -- co-running kernels are time-sliced, so their
#     durations are inflated: read per-kernel figures from (2)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_ms -o bench -- python3 bench.py --steps 6 --warmup 2 --no_cpu_baseline --resident --no_runner --no_native_leg > $OUT/bench_trace_ms.json 2> $OUT/trace_ms.err
# (2) every side stream off: per-kernel durations are those of kernels running alone (bench.py measures its roofline step
#     the same way); the PMC passes do not depend on it
export HND_TEACHER_STREAM=0
export HND_DEFER_FPN=0
export HND_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --steps 6 --warmup 2 --no_cpu_baseline --resident --no_runner --no_native_leg > $OUT/bench_trace.json 2> $OUT/trace.err
# PMC passes: counters only with --kernel-trace (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: separate passes)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python3 bench.py --steps 1 --warmup 1 --no_cpu_baseline --resident --no_runner --no_native_leg > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python3 bench.py --steps 1 --warmup 1 --no_cpu_baseline --resident --no_runner --no_native_leg > $OUT/bench_write.json 2> $OUT/write.err
# effective clock under each kernel family (DVFS: the chip lowers its clock under load): GRBM_GUI_ACTIVE / 8 / wall
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_clock -o bench -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --resident --no_runner --no_native_leg > $OUT/bench_clock.json 2> $OUT/clock.err
python3 tools/effective_clock.py $OUT/pmc_clock "default step (covered GEMMs fp32-emulated on the bf16 pipe)" > $OUT/effective_clock.md 2>> $OUT/clock.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_clock_bx3 -o bench -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --resident --no_runner --native_fp32 > $OUT/bench_clock_bx3.json 2>> $OUT/clock.err
python3 tools/effective_clock.py $OUT/pmc_clock_bx3 "HND_BF16X3=0 step (native fp32 MFMA everywhere)" >> $OUT/effective_clock.md 2>> $OUT/clock.err
rm -f $OUT/pmc_clock/*kernel_trace.csv $OUT/pmc_clock_bx3/*kernel_trace.csv
find $OUT/pmc_clock $OUT/pmc_clock_bx3 -name "*counter_collection.csv" -delete
# neural filter training step, batch 16: kernel stats only
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/filter -o filter -- python3 tools/bench_filter.py --batch 16 --steps 10 --cpu_steps 0 > $OUT/bench_filter.json 2> $OUT/filter.err
rm -f $OUT/filter/filter_kernel_trace.csv
python3 tools/summarize_profile.py $OUT $R > $OUT/summary.md 2> $OUT/summary.err
ls -la $OUT $OUT/* | head -40
# keep the merged-back payload small
python3 tools/idle_gaps.py $OUT/trace_ms/bench_kernel_trace.csv > $OUT/idle_gaps.txt 2>&1
rm -f $OUT/trace/bench_kernel_trace.csv $OUT/trace_ms/bench_kernel_trace.csv $OUT/pmc_fetch/bench_kernel_trace.csv $OUT/pmc_write/bench_kernel_trace.csv
find $OUT -name "*counter_collection.csv" -size +20M -delete
