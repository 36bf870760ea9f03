#!/bin/bash
# Sample the shader clock / power while a command runs: tools/sample_clocks.sh <out.txt> -- <cmd...>
# (a measurement aid: the fp32-MFMA peak in DESIGN.md is quoted at 2.4 GHz; this shows what the part holds under load)
out=$1; shift; shift
( while true; do
    /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo
    sleep 0.2
  done ) > "$out" &
sampler=$!
"$@"
rc=$?
kill $sampler 2>/dev/null
exit $rc
