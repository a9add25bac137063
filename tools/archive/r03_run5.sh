#!/bin/bash
# A/B of the fused masked gather in D's pooled backward: kernel stats of the micro-benchmark, then the bench line both ways.
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r03k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
GB_N=32 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gb -o gb -- python3 $R/tools/gather_bwd_bench.py > $O/prof_gb.log 2>&1 || exit 1
cd $R
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > $O/bench_gather.json 2> $O/bench_gather.err || exit 1
SARAGAN_NO_GATHER_BWD=1 timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > $O/bench_planes.json 2> $O/bench_planes.err || exit 1
tail -c 300 $O/bench_gather.json; tail -c 300 $O/bench_planes.json
