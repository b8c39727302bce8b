import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from score_amd.manhattan import make_manhattan
from score_amd.refine import refine_estimate
from test_refine import _noisy_truth
fg = make_manhattan(n_robots=20, n_poses=1000, n_beacons=4, seed=3000)
res = _noisy_truth(fg)
for i in range(3):
    t = time.perf_counter(); out, info = refine_estimate(fg, res, solver_settings=dict(verbose=1 if i == 2 else 0)); dt = time.perf_counter() - t
    print(f"refine_estimate {dt*1e3:.1f} ms: create {info['setup_ms']:.1f} run {info['solve_ms']:.1f} its {info['iterations']} pcg {info['pcg_iters']}", flush=True)
