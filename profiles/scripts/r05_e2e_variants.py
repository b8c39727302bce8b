"""Why the bench's from-arrays sweeps (47 ms) are slower than the timeline script's (37 ms): the same 64 ArrayGraphs through
solve_score_batch with (a) the object graphs alive or not, (b) the garbage collector on / frozen / off.
python profiles/scripts/r05_e2e_variants.py"""
import gc, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from score_amd.manhattan import make_manhattan
from score_amd.native import ArrayGraph, graph_arrays
from score_amd.solve_score import solve_score_batch

st = dict(device=0, eps_abs=1e-7, eps_rel=1e-7)
trials = [make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=5000 + t) for t in range(64)]
flat = [ArrayGraph(graph_arrays(fg)) for fg in trials]

def sweeps(label, n=6):
    solve_score_batch(flat, "SOCP", solver_settings=st, workers=8)
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); solve_score_batch(flat, "SOCP", solver_settings=st, workers=8); ts.append(1e3 * (time.perf_counter() - t0))
    print(f"{label:46s} " + " ".join(f"{t:5.1f}" for t in ts) + f"  median {sorted(ts)[len(ts)//2]:.1f} ms = {64e3/sorted(ts)[len(ts)//2]:.0f}/s", flush=True)

sweeps("objects alive, gc on")
gc.collect(); gc.freeze()
sweeps("objects alive, gc frozen")
gc.unfreeze(); gc.disable()
sweeps("objects alive, gc disabled")
gc.enable()
del trials
gc.collect()
sweeps("objects dropped, gc on")
solve_score_batch([make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=7000 + t) for t in range(16)], "SOCP", solver_settings=st, workers=8)
sweeps("after a from-objects batch, gc on")
