#!/bin/bash
# Round-5 evidence (one gpurun call): counter passes over the hot shapes (ONE counter group per pass, --kernel-trace only, as
# MI355X_MICROARCH.md prescribes), the kernel-stats trace of the bench command, the clock the isolated PMC launches ran at.
# Results under gpurun_out/final/; `PROFILE_TAG=r05_ python tools/refresh_profiles.py` copies them into profiles/.
R="$GRAFT_REPO_ROOT"
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { local t=$1; shift; timeout -k 10 "$t" "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return $rc; }
( while true; do echo "T $(date +%s.%N) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | sed -E 's/.*: //' | tr '\n' '|')"; sleep 0.5; done ) > $O/pmc_clock_samples.txt &
SMI=$!
echo "pmc_rd begins $(date +%s.%N)" > $O/pmc_pass_times.txt
run 250 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_rd -o rd -- python3 $R/tools/pmc_probe.py > $O/pmc_rd.log 2>&1 || { kill $SMI; exit 1; }
echo "pmc_wr begins $(date +%s.%N)" >> $O/pmc_pass_times.txt
run 250 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_wr -o wr -- python3 $R/tools/pmc_probe.py > $O/pmc_wr.log 2>&1 || { kill $SMI; exit 1; }
echo "pmc_mfma begins $(date +%s.%N)" >> $O/pmc_pass_times.txt
run 250 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -o mf -- python3 $R/tools/pmc_probe.py > $O/pmc_mf.log 2>&1 || { kill $SMI; exit 1; }
echo "pmc_sq begins $(date +%s.%N)" >> $O/pmc_pass_times.txt
run 250 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -o sq -- python3 $R/tools/pmc_probe.py > $O/pmc_sq.log 2>&1 || { kill $SMI; exit 1; }
echo "pmc passes end $(date +%s.%N)" >> $O/pmc_pass_times.txt
kill $SMI
cp $R/gpurun_out/pmc_manifest.json $O/pmc_manifest.json
run 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/prof_bench.log 2>&1 || exit 1
cd $R
run 300 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
run 200 python bench.py --dump-prof --steps 5 --warmup 3 --no-extras --no-cpu-baseline > /dev/null 2> $O/bench_conv_table.txt
for c in 1 2 4 5; do run 200 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_config$c.json 2> $O/bench_config$c.err; done
# "stabilising phase (alpha = 0) plus one mixing run (alpha = 0.5, freeze ops)" of the headline workload
run 300 python bench.py --alpha 0.5 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_mixing.json 2> $O/bench_mixing.err
# the reference's own operating point (out.txt: xs phase 5, 64x64x16) at local batch 2 / 4 / 8, and the headline network at 2 / 4
: > $O/bench_reference_point.jsonl
for b in 2 4 8; do run 200 python bench.py --config out_txt --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2> $O/bench_ref_b$b.err | grep '^{' >> $O/bench_reference_point.jsonl; done
for b in 2 4; do run 200 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2> $O/bench_cfg3_b$b.err | grep '^{' >> $O/bench_reference_point.jsonl; done
# three more default lines: the spread between launches on one box
: > $O/bench_repeats.jsonl
for i in 1 2 3; do run 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' >> $O/bench_repeats.jsonl; done
ls $O
