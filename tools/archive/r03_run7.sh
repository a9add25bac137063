#!/bin/bash
# First process on a fresh box vs the second: per-step device / host times of the timed region.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03r; mkdir -p $O
for i in 1 2 3; do
  SARAGAN_BENCH_STEP_TIMES=1 timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline > $O/run$i.json 2> $O/run$i.err || exit 1
  python -c "
import json
d=json.loads(open('$O/run$i.json').read().strip().splitlines()[-1]); print('run$i', d['value'], d['ms_per_step'])"
  grep STEP_TIMES $O/run$i.err
done
