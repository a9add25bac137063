#!/bin/bash
# round 3, GPU call 1: the whole GPU suite (new parity tests included), the default bench line with its table.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03a
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 "$t" "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return $rc; }
run 900 python -m pytest tests -m gpu -q -rf --durations=15 > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -40 $O/pytest_gpu.log
run 500 python bench.py --dump-prof > $O/bench_default.json 2> $O/bench_conv_table.txt; tail -c 2500 $O/bench_default.json
run 300 python bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline --dump-prof > $O/bench_config5.json 2> $O/bench_config5.err; tail -c 600 $O/bench_config5.json
