#!/bin/bash
# Round-6 profiler artefacts (run through gpurun from the repo root):   bash profiles/scripts/r06_profile.sh <tag>
# 1. kernel trace of the LOOP ONLY (timed region of bench.py, no probes): the average duration of k_spmv_band<1, 2, 4, true> here is
#    what roofline.us_per_launch must agree with;  2. kernel trace of the default bench command (all legs, no CPU baseline);
# 3. PMC passes FETCH_SIZE / WRITE_SIZE, one counter per run, kernel trace only: the ADMM loop of the headline problem, of a
#    batch of 16, and the default (Newton) solves of the headline problem and of a 16-trial config-5 handle.
set -u
TAG=${1:-r06}
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
# (round 6: the batch-16 loop's kernel trace -- the duration behind roofline_batch16, which round 5 took from HIP events alone)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/batch16" -o b16 --output-format csv -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-probes --batch 16 > "$OUT/batch16_bench.json" 2> "$OUT/batch16.err"
# (... and a graph with loop closures through the default solver: the link kernels beside the chain kernel)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/links" -o lk --output-format csv -- python3 "$REPO/profiles/scripts/r06_links_workload.py" > "$OUT/links_workload.txt" 2> "$OUT/links.err"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_headline_loop_$C" -o p --output-format csv -- python3 "$REPO/bench.py" --steps 1 --warmup 0 --no-probes > /dev/null 2> "$OUT/pmc1_$C.err"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_batch16_loop_$C" -o p --output-format csv -- python3 "$REPO/bench.py" --steps 1 --warmup 0 --no-probes --batch 16 > /dev/null 2> "$OUT/pmc16_$C.err"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_newton_headline_$C" -o p --output-format csv -- python3 "$REPO/profiles/scripts/r04_newton_workload.py" headline > /dev/null 2> "$OUT/pmcn_$C.err"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_newton_mc16_$C" -o p --output-format csv -- python3 "$REPO/profiles/scripts/r04_newton_workload.py" mc16 > /dev/null 2> "$OUT/pmcm_$C.err"
done
cd "$REPO"
# (the tracked summaries are written by profiles/scripts/r04_summarise.py gpurun_out/<tag> <tag>, run where the repository is:
#  only gpurun_out/ travels back from the GPU box)
# ---- round 5: the fresh-graph Monte-Carlo sweep (trace + busy fraction), one default solve (gaps), setup phases, timelines ----
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT/fresh" -o f164 -- python3 "$REPO/profiles/scripts/r05_fresh_trace.py" 16 4 5 > "$OUT/fresh_f164.txt" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT/solve" -o hl -- python3 "$REPO/profiles/scripts/r05_solve_trace.py" 6 > "$OUT/solve_hl.txt" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/create" -o b16 -- python3 "$REPO/profiles/scripts/r05_create_trace.py" 4 16 6 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/create" -o hl -- python3 "$REPO/profiles/scripts/r05_create_trace.py" 20 1 6 > /dev/null 2>&1
cd "$REPO"
python3 profiles/scripts/r05_solve_gaps.py "$OUT/solve/hl_results.db" > "$OUT/default_solve_gaps.txt" 2>&1
python3 - "$OUT" <<'PY' > "$OUT/montecarlo_trace.txt" 2>&1
import sqlite3, subprocess, sys, re
out = sys.argv[1]
last = [l for l in open(out + "/fresh_f164.txt") if l.startswith("SWEEP")][-1].split()
dur = float(last[4]) * 1e6
con = sqlite3.connect(out + "/fresh/f164_results.db"); mx = con.execute("select max(end) from kernels").fetchone()[0]
print("last sweep of r05_fresh_trace.py 16 4 5 (64 fresh graphs: 4 lock-step handles of 16 on 4 host threads):", last[4], "ms")
print(subprocess.run([sys.executable, "profiles/scripts/r05_trace_busy.py", out + "/fresh/f164_results.db", str(int(mx + 3e5 - dur)), str(int(mx + 3e5))], capture_output=True, text=True).stdout)
PY
python3 - "$OUT" <<'PY' > "$OUT/create_kernels.txt" 2>&1
import sqlite3, sys
for f, what in (("b16", "16 x (4 robots x 1000 poses)"), ("hl", "1 x (20 robots x 1000 poses)")):
    con = sqlite3.connect(sys.argv[1] + "/create/" + f + "_results.db"); cur = con.cursor()
    rows = cur.execute("select name, count(*), sum(end-start)/1000.0, avg(end-start)/1000.0 from kernels group by name order by 3 desc limit 24").fetchall()
    tot = cur.execute("select sum(end-start)/1000.0, count(*) from kernels").fetchone()
    print("== score_create_from_graphs,", what, ": %.1f us of kernels per create, %d launches" % (tot[0] / 6, tot[1] / 6))
    for r in rows: print(f"{r[2]/6:9.1f} us/create {r[1]//6:4d} x {r[3]:8.2f} us  {r[0][:100]}")
PY
for cfg in "10 20 1" "8 4 16"; do timeout 300 python3 profiles/scripts/r05_create.py $cfg 2>&1 | grep -v "destroy\|context sw" >> "$OUT/setup.txt"; done
timeout 300 python3 profiles/scripts/r05_e2e_timeline.py 8 8 > "$OUT/e2e_timeline.txt" 2>&1
timeout 300 python3 profiles/scripts/r05_fresh.py 5 > "$OUT/fresh_sweeps.txt" 2>&1
timeout 300 python3 profiles/scripts/r05_mc_occ2.py 10 2>&1 | grep "OCC2=0" > "$OUT/mc_resolve.txt"
# (back home: for f in default_solve_gaps montecarlo_trace create_kernels setup e2e_timeline fresh_sweeps mc_resolve; do cp gpurun_out/<tag>/$f.txt profiles/<tag>_$f.txt; done)
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
du -sh "$OUT"
