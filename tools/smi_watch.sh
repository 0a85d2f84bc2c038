#!/bin/bash
# Samples clocks and power once a second while a command runs: tools/smi_watch.sh <out.txt> <command...>
out=$1; shift
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr -s ' ' | tr '\n' ';'; echo; sleep 1; done ) > $out &
w=$!
"$@"
rc=$?
kill $w
exit $rc
