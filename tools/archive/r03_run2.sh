#!/bin/bash
# round 3, GPU call 2: why do the legs after the main loop run faster?  One process under rocprofv3 --kernel-trace with
# marker kernels around every timed region (tools/trace_windows.py), an A/B without any event bracketing, and the printed
# reports of the bf16-emulation tests.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03b
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 "$t" "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return $rc; }
export SARAGAN_BENCH_MARK=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --no-cpu-baseline --steps 10 > $O/bench_traced.json 2> $O/bench_traced.err
f=$(ls $O/trace/*/*kernel_trace.csv | head -1); echo "trace: $f $(wc -l < $f) rows"
python tools/trace_windows.py "$f" > $O/leg_windows.txt 2>&1; head -60 $O/leg_windows.txt
gzip -9 -c "$f" > $O/kernel_trace.csv.gz; rm -rf $O/trace
unset SARAGAN_BENCH_MARK
SARAGAN_BENCH_NO_PROF=1 run 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_noprof.json 2> $O/bench_noprof.err; python - <<'PY'
import json
for f in ('bench_traced','bench_noprof'):
    try:
        r=[json.loads(l) for l in open(f'gpurun_out/r03b/{f}.json') if l.startswith('{')][-1]
        print(f, r['ms_per_step'], {k:v['ms_per_step'] for k,v in r.get('extras',{}).items()})
    except Exception as e: print(f, 'ERR', e)
PY
run 600 python -m pytest tests/test_step_gpu.py tests/test_configs_gpu.py tests/test_networks2d_gpu.py -q -s -k "emulating or config2 or config5_full" > $O/emu_tests.log 2>&1; echo "rc=$?"; grep -E "vs bf16|oracle_step|passed|failed" $O/emu_tests.log | cut -c1-1800
