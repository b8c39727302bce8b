#!/bin/bash
# kernel trace of BASELINE configs[4] on one GPU (3 sweeps of 64 trials): which kernels own the GPU, how busy is it
set -u
OUT=$PWD/gpurun_out/mc_trace
mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT" -o mc --output-format csv -- python3 "$REPO/bench.py" --workload montecarlo --steps 3 --warmup 1 --no-probes > "$OUT/bench.json" 2> "$OUT/err.txt"
cd "$REPO"
python3 - <<'PY'
import csv, glob, json
f = glob.glob("gpurun_out/mc_trace/**/mc_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# the timed region: the last 3/4 of the launches roughly; take the busiest contiguous window = everything after warm-up
n = len(ev)
ev = ev[n // 4:]
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]; ksum = 0
for s, e, _ in ev:
    ksum += e - s
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"window {1e-6*(t1-t0):.1f} ms: GPU busy (union of kernel intervals) {100*busy/(t1-t0):.0f} %, sum of kernel durations / window = {ksum/(t1-t0):.2f} (overlap between streams)")
agg = {}
for s, e, k in ev:
    k = k.split("(")[0].replace("void score::", "").replace("score::", "")
    a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {k:36s} {c:7d} launches  {1e-6*d:8.2f} ms  avg {1e-3*d/c:7.2f} us  {100*d/ksum:5.1f} %")
print(open("gpurun_out/mc_trace/bench.json").read()[:300])
PY
