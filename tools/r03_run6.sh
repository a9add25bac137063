#!/bin/bash
# Does one large up-front allocation change the step?  (the later legs of a bench process ran faster on some boxes)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03q; mkdir -p $O
for i in 1 2; do
  timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline > $O/plain$i.json 2>/dev/null || exit 1
  SARAGAN_ARENA_GB=96 timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline > $O/arena$i.json 2>/dev/null || exit 1
done
PYTORCH_HIP_ALLOC_CONF=expandable_segments:True timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline > $O/expand.json 2>$O/expand.err
for f in plain1 arena1 plain2 arena2 expand; do python -c "
import json,sys
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['avg_ms'])"; done
