"""64 config-4 trials through solve_score_batch from graph OBJECTS: where the difference to the flat-array path goes.
python profiles/scripts/r04_e2e_objects.py"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import score_amd.solve_score as S
from score_amd.manhattan import make_manhattan
from score_amd.native import ArrayGraph, graph_arrays
trials = [make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=5000 + t) for t in range(64)]
st = dict(device=0)
def best(f, n=5):
    f(); ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return 1e3 * min(ts), 1e3 * sorted(ts)[n // 2]
print("graph_arrays, 64 graphs, one thread: min %.1f ms median %.1f" % best(lambda: [graph_arrays(fg) for fg in trials]))
flat = [ArrayGraph(graph_arrays(fg)) for fg in trials]
print("sweep from arrays : min %.1f ms median %.1f" % best(lambda: S.solve_score_batch(flat, "SOCP", solver_settings=st, workers=8)))
print("sweep from objects: min %.1f ms median %.1f" % best(lambda: S.solve_score_batch(trials, "SOCP", solver_settings=st, workers=8)))
old = sys.getswitchinterval()
for si in (1e-3, 2e-4):
    sys.setswitchinterval(si)
    print("  switch interval %.4f: objects min %.1f ms median %.1f" % ((si,) + best(lambda: S.solve_score_batch(trials, "SOCP", solver_settings=st, workers=8))))
sys.setswitchinterval(old)
def pre():
    fl = [ArrayGraph(graph_arrays(fg)) for fg in trials]
    return S.solve_score_batch(fl, "SOCP", solver_settings=st, workers=8)
print("arrays first (calling thread), then the sweep: min %.1f ms median %.1f" % best(pre))
