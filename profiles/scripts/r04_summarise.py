#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of r04_profile.sh into the small tracked summaries under profiles/.
    python3 profiles/scripts/r04_summarise.py gpurun_out/<tag> <tag>
* <tag>_loop_only_kernel_stats.csv / <tag>_full_kernel_stats.csv  -- rocprofv3 --kernel-trace --stats
* <tag>_pmc_fetch_write_per_kernel.json -- per workload section (headline_loop, batch16_loop, newton_headline, newton_mc16) the
  mean FETCH_SIZE / WRITE_SIZE per kernel name (KB, as rocprofv3 reports them), from separate --pmc passes, and the
  gfx950-corrected HBM bytes per launch (FETCH_SIZE x 2: exact for 16-byte lane loads, an upper bound for 8- / 4-byte ones --
  /opt/skills/guides/MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
prof = os.path.join(root, "profiles")


def find(sub, suffix):
    hits = glob.glob(os.path.join(out, sub, "**", f"*{suffix}"), recursive=True)
    return hits[0] if hits else None


for sub, name in (("loop", f"{tag}_loop_only_kernel_stats.csv"), ("full", f"{tag}_full_kernel_stats.csv"),
                  ("batch16", f"{tag}_batch16_loop_kernel_stats.csv"), ("links", f"{tag}_links_kernel_stats.csv")):
    src = find(sub, "kernel_stats.csv")
    if src:
        shutil.copy(src, os.path.join(prof, name)); print("wrote", name)
commands = {"headline_loop": "bench.py --steps 1 --warmup 0 --no-probes", "batch16_loop": "bench.py --steps 1 --warmup 0 --no-probes --batch 16",
            "newton_headline": "profiles/scripts/r04_newton_workload.py headline", "newton_mc16": "profiles/scripts/r04_newton_workload.py mc16"}
rec = {"command": "rocprofv3 --kernel-trace --pmc <COUNTER> -- python3 <section command> (one pass per counter)",
       "unit": "KB per launch as reported by rocprofv3", "sections": {}}
for sec, cmd in commands.items():
    counters = {}
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        src = find(f"pmc_{sec}_{cname}", "counter_collection.csv")
        if not src:
            continue
        acc = defaultdict(lambda: [0.0, 0])
        with open(src) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != cname:
                    continue
                a = acc[row["Kernel_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
        counters[cname] = {k: {"mean_KB": v[0] / v[1], "launches": v[1]} for k, v in acc.items() if "score::" in k}
    if not counters:
        continue
    s = {"command": cmd, "counters": counters, "hbm_bytes_per_launch": {}}
    for k in counters.get("FETCH_SIZE", {}):
        f = counters["FETCH_SIZE"][k]["mean_KB"] * 1024.0
        w = counters.get("WRITE_SIZE", {}).get(k, {"mean_KB": 0.0})["mean_KB"] * 1024.0
        s["hbm_bytes_per_launch"][k] = {"fetch_raw": f, "fetch_gfx950_corrected_upper_bound": 2.0 * f, "write": w, "traffic_upper_bound": 2.0 * f + w}
    rec["sections"][sec] = s
if rec["sections"]:
    with open(os.path.join(prof, f"{tag}_pmc_fetch_write_per_kernel.json"), "w") as fh:
        json.dump(rec, fh, indent=1, sort_keys=True)
    print("wrote", f"{tag}_pmc_fetch_write_per_kernel.json", list(rec["sections"]))
for sub in ("loop", "full", "batch16"):
    src = os.path.join(out, f"{sub}_bench.json")
    if os.path.exists(src) and os.path.getsize(src):
        shutil.copy(src, os.path.join(prof, f"{tag}_{sub}_bench.json"))
