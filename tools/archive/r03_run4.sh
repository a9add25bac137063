#!/bin/bash
# round 3, GPU call: full GPU suite on the current build + kernel stats of the default bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03g
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 "$t" "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return $rc; }
run 900 python -m pytest tests -m gpu -q -rf > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -12 $O/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/bench_prof.json 2> $O/bench_prof.err
f=$(ls $O/prof/*/*kernel_stats.csv | head -1); cp $f $O/kernel_stats.csv; rm -rf $O/prof; head -40 $O/kernel_stats.csv | cut -c1-170
