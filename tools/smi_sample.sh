#!/bin/bash
# Samples GPU clock / power with rocm-smi while a command runs: tools/smi_sample.sh out.log -- cmd args...
out=$1; shift; shift
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk|fclk" | tr -s ' ' | tr '\n' '|' ; echo; sleep 0.25; done ) > "$out" &
spid=$!
"$@"
rc=$?
kill $spid
exit $rc
