"""A/B of the chain kernel compiled for two workgroups per CU (SCORE_PREC_OCC2=1: 128 registers, spills) in the Monte-Carlo
re-solve regime: 64 config-5 trials in 4 lock-step handles of 16, one host thread each.  python r05_mc_occ2.py [sweeps]"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from score_amd.manhattan import make_manhattan
from score_amd.native import assemble_native
from score_amd.solver import ConicSolver

sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
qps = [assemble_native(make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=4000 + t), "SOCP").qp for t in range(64)]
for mode in ("0", "1", "0", "1"):
    if mode == "1": os.environ["SCORE_PREC_OCC2"] = "1"
    else: os.environ.pop("SCORE_PREC_OCC2", None)
    # (the switch is read once per process: run each mode in a child)
    import subprocess
    code = f"""
import os, sys, time
sys.path.insert(0, {ROOT!r})
from concurrent.futures import ThreadPoolExecutor
from score_amd.manhattan import make_manhattan
from score_amd.native import assemble_native
from score_amd.solver import ConicSolver
qps = [assemble_native(make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=4000 + t), "SOCP").qp for t in range(64)]
hs = [ConicSolver(qps[i:i+16], {{}}) for i in range(0, 64, 16)]
with ThreadPoolExecutor(4) as pool:
    for _ in range(3): list(pool.map(lambda h: h.solve(), hs))
    ts = []
    for _ in range({sweeps}):
        t0 = time.perf_counter(); rs = list(pool.map(lambda h: h.solve(), hs)); ts.append(time.perf_counter() - t0)
ts.sort()
print("OCC2=" + os.environ.get("SCORE_PREC_OCC2", "0"), "median sweep %.2f ms = %.0f problems/s, best %.2f ms, solved %d" % (1e3*ts[len(ts)//2], 64/ts[len(ts)//2], 1e3*ts[0], sum(r.solved for x in rs for r in x)), flush=True)
"""
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ))
