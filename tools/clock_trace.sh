#!/bin/bash
# Diagnostic: samples the GPU's clocks and power (rocm-smi) every ~0.25 s while `python bench.py` runs; bench prints leg
# markers with SARAGAN_BENCH_MARK=1.  usage: bash tools/clock_trace.sh > gpurun_out/clock_trace.txt
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
( while true; do echo "T $(date +%s.%N) $(rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E 'sclk|mclk|Power|Temperature \(Sensor junction\)' | tr -s ' ' | tr '\n' '|')"; sleep 0.25; done ) &
SMI=$!
SARAGAN_BENCH_MARK=1 timeout -k 10 300 python bench.py --no-cpu-baseline 2>&1 | grep -E "^MARK|^\{" | cut -c1-400
kill $SMI
