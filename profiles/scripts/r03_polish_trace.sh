#!/bin/bash
# kernel timeline of product-default solves of the headline problem (summary of the last solve)
set -u
OUT=$PWD/gpurun_out/polish_trace
mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT" -o pt --output-format csv -- python3 "$REPO/profiles/scripts/r02_trace_polish.py" > "$OUT/out.txt" 2> "$OUT/err.txt"
cd "$REPO"
cat "$OUT/out.txt"
python3 profiles/scripts/r02_timeline.py "$(find gpurun_out/polish_trace -name 'pt_kernel_trace.csv' | head -1)"
