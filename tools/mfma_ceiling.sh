#!/bin/bash
# the sustained bf16 MFMA rate of THIS board with its clock, power and power cap beside it.
# usage (on the GPU box): bash tools/mfma_ceiling.sh > gpurun_out/mfma_ceiling.txt 2>&1
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
[ -x tools/probe/mfma_ceiling ] || hipcc --offload-arch=gfx950 -O3 -o tools/probe/mfma_ceiling tools/probe/mfma_ceiling.hip || exit 1
echo "== board power cap"
rocm-smi --showmaxpower 2>/dev/null | grep -E "Max Graphics Package Power|GPU\[" 
for f in /sys/class/drm/card*/device/hwmon/hwmon*/power1_cap /sys/class/drm/card*/device/hwmon/hwmon*/power1_cap_max /sys/class/drm/card*/device/hwmon/hwmon*/power1_cap_default; do
  [ -r "$f" ] && echo "$f = $(cat $f) uW"
done
rocm-smi --showperflevel --showsclkrange 2>/dev/null | grep -E "GPU\[" 
echo "== samples (every ~0.3 s): epoch | junction C | sclk | power W"
( while true; do echo "T $(date +%s.%N) $(rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E 'sclk|Power|Temperature \(Sensor junction\)' | sed -E 's/.*: //' | tr '\n' '|')"; sleep 0.25; done ) > gpurun_out/mfma_ceiling_smi.txt &
SMI=$!
echo "== run (starts $(date +%s.%N))"
timeout -k 10 120 tools/probe/mfma_ceiling ${1:-6}
rc=$?
echo "== run ends $(date +%s.%N) rc=$rc"
kill $SMI
echo "== rocm-smi samples during the run"
cat gpurun_out/mfma_ceiling_smi.txt
exit $rc
