#!/bin/bash
# Run on the GPU box (via gpurun): only the effective-clock passes of tools/profile_round.sh.  usage: tools/profile_clock.sh r05
set -u
R=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$R
mkdir -p $OUT
export HND_TEACHER_STREAM=0
export HND_DEFER_FPN=0
export HND_WGRAD_STREAM=0
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_clock -o bench -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --resident --no_runner > $OUT/bench_clock.json 2> $OUT/clock.err
ls -la $OUT/pmc_clock | head; head -c 600 $OUT/pmc_clock/*counter_collection.csv
python3 tools/effective_clock.py $OUT/pmc_clock "native fp32 MFMA step" > $OUT/effective_clock.md 2>> $OUT/clock.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_clock_bx3 -o bench -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --resident --no_runner --bf16x3 > $OUT/bench_clock_bx3.json 2>> $OUT/clock.err
python3 tools/effective_clock.py $OUT/pmc_clock_bx3 "HND_BF16X3=1 step (opt-in proposal)" >> $OUT/effective_clock.md 2>> $OUT/clock.err
rm -f $OUT/pmc_clock/*kernel_trace.csv $OUT/pmc_clock_bx3/*kernel_trace.csv
find $OUT/pmc_clock $OUT/pmc_clock_bx3 -name "*counter_collection.csv" -delete
cat $OUT/effective_clock.md; tail -5 $OUT/clock.err
