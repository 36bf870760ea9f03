#!/bin/bash
# Run on the GPU box (via gpurun): SQ counters (matrix-pipe busy, wave parked / issue-stalled) of the persistent GEMM
# kernels on the step's shapes.  Counters only -- no sys / hip / memory traces in the same pass (gpurun refuses those).
# usage: tools/profile_sq.sh r04
set -u
R=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/sq_$R
mkdir -p $OUT
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_BUSY_CYCLES"
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/bres -o bres -- python3 tools/bench_bres.py --iters 3 > $OUT/bres.log 2> $OUT/bres.err
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/bstream -o bstream -- python3 tools/bench_bstream.py --iters 3 > $OUT/bstream.log 2> $OUT/bstream.err
# the emulated family: the B-resident and the B-streamed emulation kernels beside the native picks
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/bx3 -o bx3 -- python3 tools/probes/bx3_shape_bench.py > $OUT/bx3.log 2> $OUT/bx3.err
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/bxs -o bxs -- python3 tools/bench_bxs.py > $OUT/bxs.log 2> $OUT/bxs.err
python3 tools/sq_counters.py $OUT/bx3 "fp32 emulated on the bf16 pipe, B resident (tools/probes/bx3_shape_bench.py shapes, batch 16) beside the native kernels" > $OUT/${R}_bx3_sq_counters.txt 2>> $OUT/sum.err
python3 tools/sq_counters.py $OUT/bxs "fp32 emulated on the bf16 pipe, B streamed (tools/bench_bxs.py shapes, batch 16) beside the native kernels" > $OUT/${R}_bxs_sq_counters.txt 2>> $OUT/sum.err
python3 tools/sq_counters.py $OUT/bres "B-resident persistent GEMMs (tools/bench_bres.py shapes, batch 16): bres2 = one wave per SIMD, bres = 8 waves, igemm = tiled" > $OUT/${R}_bres2_sq_counters.txt 2> $OUT/sum.err
python3 tools/sq_counters.py $OUT/bstream "B-streamed persistent GEMM (tools/bench_bstream.py shapes, batch 16)" > $OUT/${R}_bstream_sq_counters.txt 2>> $OUT/sum.err
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
ls -la $OUT
