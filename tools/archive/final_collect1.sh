#!/bin/bash
# Round-end evidence, part 1 (one gpurun call): the full GPU suite, the default bench line (+ per-shape table), the other
# BASELINE configurations.  A step that hits its timeout ends the script (no GPU step after a killed one).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/final
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 "$t" "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return $rc; }
run 800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
run 500 python bench.py --dump-prof > $O/bench_default.json 2> $O/bench_conv_table.txt; tail -c 1500 $O/bench_default.json
for c in 1 2 4 5; do
  run 300 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_config$c.json 2> $O/bench_config$c.err; echo "config $c rc=$?"; tail -c 400 $O/bench_config$c.json
done
# the N > 1 launcher rehearsed on this one-GPU box: two ranks stacked on the card, gloo collectives, a small batch
SARAGAN_BENCH_STACK_RANKS=1 SARAGAN_DIST_BACKEND=gloo run 400 python bench.py --gpus 2 --steps 3 --warmup 2 --batch 8 --no-extras --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo2 rc=$?"; tail -c 600 $O/bench_gloo2.json
