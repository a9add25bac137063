#!/bin/bash
# Round-end evidence, part 2 (one gpurun call): counter passes over the hot shapes (one counter group per pass), the
# kernel-stats trace of the bench command, elementwise roofline, probes, phase stamps.
R="$GRAFT_REPO_ROOT"
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { local t=$1; shift; timeout -k 10 "$t" "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return $rc; }
run 250 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_rd -o rd -- python3 $R/tools/pmc_probe.py > $O/pmc_rd.log 2>&1 || exit 1
run 250 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_wr -o wr -- python3 $R/tools/pmc_probe.py > $O/pmc_wr.log 2>&1 || exit 1
run 250 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -o mf -- python3 $R/tools/pmc_probe.py > $O/pmc_mf.log 2>&1 || exit 1
run 250 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -o sq -- python3 $R/tools/pmc_probe.py > $O/pmc_sq.log 2>&1 || exit 1
cp $R/gpurun_out/pmc_manifest.json $O/pmc_manifest.json
run 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/prof_bench.log 2>&1 || exit 1
cd $R
run 200 python tools/ew_roofline.py --json $O/elementwise_roofline.json > $O/elementwise_roofline.txt 2>&1
run 60 ./saragan_amd/build/probe/store_war_probe > $O/store_war_probe.txt 2>&1
run 60 ./saragan_amd/build/probe/mfma_lds_probe > $O/mfma_lds_probe.txt 2>&1
run 60 ./saragan_amd/build/probe/pk_f32_probe > $O/pk_f32_probe.txt 2>&1
{ echo "== conv_fwd3s n32 32->64 128^2"; TS_N=32 TS_CIN=32 TS_COUT=64 run 100 python tools/ts_conv.py;
  echo "== conv_fwd3s masked epilogue, fine stamps (SG_DBG_FLAGS=128)"; SG_DBG_FLAGS=128 TS_EPI=mask TS_N=32 TS_CIN=32 TS_COUT=64 run 100 python tools/ts_conv.py;
  echo "== conv_fwd5 n32 64->128 64^2"; TS_N=32 TS_DHW=16,64,64 TS_CIN=64 TS_COUT=128 run 100 python tools/ts_conv.py;
  echo "== conv_wgrad3l n32 32->64 128^2"; run 100 python tools/ts_wgrad.py; } > $O/phase_stamps.txt 2>&1
ls $O
