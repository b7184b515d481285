#!/bin/bash
# polls sclk / power while a command runs: bash tools/clock_poll.sh <outfile> <cmd...>
OUT=$1; shift
( while true; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|Temperature \(Sensor (junction|edge)" | tr '\n' ' '; echo; sleep 0.25; done ) > $OUT &
P=$!
"$@"
kill $P 2>/dev/null
