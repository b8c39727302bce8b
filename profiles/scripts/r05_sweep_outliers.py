"""24 sweeps of 64 fresh graphs through solve_score_batch with the slowest create / solve+read / close of every sweep: which call
makes a slow sweep slow (round 5: close -- the block cache evicting under its old 4 GiB cap).  python r05_sweep_outliers.py [freeze]"""
import os, resource, sys, threading, time, gc
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import score_amd.solve_score as S
from score_amd.manhattan import make_manhattan
from score_amd.native import ArrayGraph, graph_arrays
from score_amd.solver import ConicSolver
log = []
def wrap(obj, name, label, cm=False):
    f = getattr(obj, name)
    f = f.__func__ if cm else f
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            log.append((label, threading.get_ident(), t, time.perf_counter()))
    setattr(obj, name, classmethod(g) if cm else g)
wrap(S, "_models_for", "models")
wrap(ConicSolver, "from_graphs", "create", cm=True)
wrap(ConicSolver, "solve_estimates", "solve+read")
wrap(ConicSolver, "close", "close")
trials = [make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=5000 + t) for t in range(64)]
flat = [ArrayGraph(graph_arrays(fg)) for fg in trials]
st = dict(device=0)
if len(sys.argv) > 1 and sys.argv[1] == "freeze":
    gc.collect(); gc.freeze()
for _ in range(2):
    S.solve_score_batch(flat, "SOCP", solver_settings=st)
for rep in range(24):
    log.clear()
    t0 = time.perf_counter()
    rs = S.solve_score_batch(flat, "SOCP", solver_settings=st)
    wall = time.perf_counter() - t0
    mx = {}
    for lab, tid, a, b in log:
        mx[lab] = max(mx.get(lab, 0.0), 1e3 * (b - a))
    infos = [r.info for r in rs]
    print(f"sweep {rep:2d}: {1e3*wall:6.1f} ms | max " + " ".join(f"{k} {v:5.1f}" for k, v in mx.items()) + f" | max setup_ms {max(i['setup_ms'] for i in infos):.1f} max solve_ms {max(i['solve_ms'] for i in infos):.1f}", flush=True)
