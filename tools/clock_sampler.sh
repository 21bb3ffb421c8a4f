#!/bin/bash
# sample sclk / power with rocm-smi while a command runs:  tools/clock_sampler.sh out.txt python bench.py ...
out=$1; shift
( while true; do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" | tr '\n' ' '; echo; sleep 0.5; done ) > "$out" &
spid=$!
"$@"
rc=$?
kill $spid
exit $rc
