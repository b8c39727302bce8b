#!/bin/bash
# Round-2 profiler artefacts (run through gpurun from the repo root):
#   bash profiles/scripts/r02_profile.sh <tag>
# 1. kernel trace of the LOOP ONLY (timed region of bench.py, no roofline probes, no extra legs):
#    the average duration of k_spmv<1> here is what roofline.us_per_launch must agree with;
# 2. kernel trace of the default bench command (all legs);
# 3. PMC passes FETCH_SIZE / WRITE_SIZE (counters in their own runs, kernel-trace only).
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/loop" -o loop --output-format csv -- python3 "$REPO/bench.py" --steps 5 --warmup 1 --no-probes > "$OUT/loop_bench.json" 2> "$OUT/loop.err"
rocprofv3 --kernel-trace --stats -d "$OUT/full" -o full --output-format csv -- python3 "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/full_bench.json" 2> "$OUT/full.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o f --output-format csv -- python3 "$REPO/bench.py" --steps 1 --warmup 0 --no-probes > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o w --output-format csv -- python3 "$REPO/bench.py" --steps 1 --warmup 0 --no-probes > /dev/null 2> "$OUT/pmc_write.err"
cd "$REPO"
python3 profiles/scripts/r02_summarise.py "$OUT" "$TAG"
