#!/bin/bash
# Diagnostic: per-step device times of a long run of the benchmarked step as the FIRST GPU process of a fresh lease, with
# rocm-smi samples (clock, power, junction / memory temperature) alongside.  usage: bash tools/transient_probe.sh [steps]
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
O=gpurun_out/transient; mkdir -p $O
( while true; do echo "T $(date +%s.%N) $(rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E 'sclk|Power|Temperature' | sed -e 's/GPU\[0\]//' | tr -s ' \t' ' ' | tr '\n' '|')"; sleep 0.5; done ) > $O/smi.txt &
SMI=$!
echo "START $(date +%s.%N)" > $O/steps.txt
SARAGAN_BENCH_NO_SETTLE=1 SARAGAN_BENCH_STEP_TIMES=1 timeout -k 10 400 python bench.py --steps ${1:-700} --warmup 3 --no-extras --no-cpu-baseline > $O/line.json 2>> $O/steps.txt
echo "END $(date +%s.%N) rc=$?" >> $O/steps.txt
kill $SMI
