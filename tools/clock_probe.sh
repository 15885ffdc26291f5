#!/bin/bash
# Sample engine clock and socket power while a command runs (evidence for the "what bounds the MFMA kernel" discussion in
# DESIGN.md).  usage: tools/clock_probe.sh <out-file> <command...>
out=$1; shift
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' ; echo; sleep 0.5; done ) > "$out" &
sampler=$!
"$@"
kill $sampler
