"""Print the figures of a bench.py JSON line that DESIGN.md quotes.  python profiles/scripts/r02_bench_summary.py <bench.json>"""
import json, sys
r = json.load(open(sys.argv[1]))
print("value", round(r["value"], 1), r["unit"], "| ms/step", round(r["ms_per_step"], 2))
p = r.get("product_default_solver")
if p:
    print("product default:", round(p["ms_per_solve"], 3), "ms/solve,", p["admm_iters"], "ADMM +", p["newton_iters"], "Newton,", p["newton_pcg_iters"], "PCG")
for k in ("roofline", "roofline_dominant", "roofline_batch16"):
    if r.get(k):
        print(k, "frac", round(r[k]["frac"], 4), "us/launch", round(r[k]["us_per_launch"], 2), "device-clock frac", round(r[k].get("frac_on_device_clock", 0), 4))
if r.get("roofline_batch16"):
    print("batch16 kernels (dispatch us):", {k: round(v, 1) for k, v in r["roofline_batch16"]["kernel_us_dispatch"].items()})
if r.get("roofline_by_kernel"):
    print("single kernels (dispatch us):", {k: round(v["us_dispatch"], 2) for k, v in r["roofline_by_kernel"].items()})
if r.get("montecarlo_64_trials_this_gpu"):
    print("config 5 on this GPU:", round(r["montecarlo_64_trials_this_gpu"]["problems_per_sec"], 1), "problems/s")
print("end_to_end:", r.get("end_to_end"))
print("speedup_time_to_solution:", r.get("speedup_time_to_solution"))
if r.get("cpu_baseline"):
    for k, v in r["cpu_baseline"]["entries"].items():
        print("  ", k, v["seconds_to_eps"], "|", v["sample"], "| cores", v["cores"])
