#!/bin/bash
# Round-4 profiler artefacts (run through gpurun from the repo root):   bash profiles/scripts/r04_profile.sh <tag>
# 1. kernel trace of the LOOP ONLY (timed region of bench.py, no probes): the average duration of k_spmv_band<1, 2, 4> here is
#    what roofline.us_per_launch must agree with;  2. kernel trace of the default bench command (all legs, no CPU baseline);
# 3. PMC passes FETCH_SIZE / WRITE_SIZE, one counter per run, kernel trace only: the ADMM loop of the headline problem, of a
#    batch of 16, and the default (Newton) solves of the headline problem and of a 16-trial config-5 handle.
set -u
TAG=${1:-r04}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/loop" -o loop --output-format csv -- python3 "$REPO/bench.py" --steps 5 --warmup 1 --no-probes > "$OUT/loop_bench.json" 2> "$OUT/loop.err"
# (the profiler crashed once in three runs of this command -- SIGSEGV inside its launch interception with four handles driven
#  from four host threads; the bench alone never did: retried once.  With the end_to_end leg -- eight host threads creating
#  handles at once -- it aborted with a malformed AQL packet and then hung in its own finalisation for the rest of the call's
#  limit: that leg is left out of the trace (--no-e2e) and every profiler command runs under `timeout`)
for try in 1 2; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/full" -o full --output-format csv -- python3 "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-e2e > "$OUT/full_bench.json" 2> "$OUT/full.err" && break
  rm -rf "$OUT/full"
done
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_headline_loop_$C" -o p --output-format csv -- python3 "$REPO/bench.py" --steps 1 --warmup 0 --no-probes > /dev/null 2> "$OUT/pmc1_$C.err"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_batch16_loop_$C" -o p --output-format csv -- python3 "$REPO/bench.py" --steps 1 --warmup 0 --no-probes --batch 16 > /dev/null 2> "$OUT/pmc16_$C.err"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_newton_headline_$C" -o p --output-format csv -- python3 "$REPO/profiles/scripts/r04_newton_workload.py" headline > /dev/null 2> "$OUT/pmcn_$C.err"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_newton_mc16_$C" -o p --output-format csv -- python3 "$REPO/profiles/scripts/r04_newton_workload.py" mc16 > /dev/null 2> "$OUT/pmcm_$C.err"
done
cd "$REPO"
python3 profiles/scripts/r04_summarise.py "$OUT" "$TAG"
