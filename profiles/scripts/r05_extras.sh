#!/bin/bash
# Round-5 text artefacts beside r05_profile.sh (run through gpurun from the repo root):   bash profiles/scripts/r05_extras.sh <tag>
# default-solve gaps, kernels of a fresh sweep / of a create / of the long-chain and 3-D loops, sweeps by grouping, end-to-end timeline.
set -u
TAG=${1:-r05}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
REPO=$PWD
S=$REPO/profiles/scripts
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT/solve" -o solve -- python3 "$S/r05_solve_trace.py" 6 > "$OUT/solve.log" 2>&1
python3 "$S/r05_solve_gaps.py" $(ls "$OUT"/solve/*/*_results.db "$OUT"/solve/*_results.db 2>/dev/null | head -1) > "$OUT/default_solve_gaps.txt" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT/fresh" -o fresh -- python3 "$S/r05_fresh_trace.py" 16 4 5 > "$OUT/fresh.log" 2>&1
W=$(grep "^SWEEP" "$OUT/fresh.log" | tail -1 | cut -d' ' -f5)
( echo "last sweep of r05_fresh_trace.py 16 4 5 (64 fresh graphs: 4 lock-step handles of 16 on 4 host threads): $W ms"; python3 "$S/r05_trace_busy.py" $(ls "$OUT"/fresh/*/*_results.db "$OUT"/fresh/*_results.db 2>/dev/null | head -1) last $W ) > "$OUT/montecarlo_trace.txt" 2>&1
for w in long 3d; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/$w" -o t --output-format csv -- python3 "$S/r05_long_trace.py" $w > "$OUT/$w.log" 2>&1
  ( grep "dispatch us" "$OUT/$w.log"; python3 - "$OUT/$w" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:12]:
    print("%-96s calls %6s avg %9.2f us  %6s %%" % (r["Name"][:96], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
  ) > "$OUT/${w}_loop_kernels.txt" 2>&1
done
cd "$REPO"
timeout 600 python3 "$S/r05_fresh.py" > "$OUT/fresh_sweeps.txt" 2>&1
timeout 600 python3 "$S/r05_e2e_timeline.py" 4 > "$OUT/e2e_timeline.txt" 2>&1
timeout 600 python3 "$S/r05_sweep_outliers.py" freeze > "$OUT/sweep_outliers.txt" 2>&1
timeout 600 python3 "$S/r05_long_solves.py" 2>&1 | grep -E "===|reset:|solve_ms" > "$OUT/long_solves.txt"
