"""Kernel-trace target: sweeps of 64 fresh graphs (score_create_from_graphs + one default solve + read-back + destroy),
groups of G on T threads; prints the wall-clock window of the LAST sweep (ns since the epoch of time.time_ns) so that the
trace can be cut to it.  python r05_fresh_trace.py [G [T [sweeps]]]"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from score_amd.manhattan import make_manhattan
from score_amd.native import graph_arrays
from score_amd.solver import ConicSolver

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
N = 64
arrs = [graph_arrays(make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=4000 + t)) for t in range(N)]
def one(idx):
    s = ConicSolver.from_graphs([arrs[i] for i in idx], 0, {})
    try:
        return s.solve_estimates()
    finally:
        s.close()
groups = [list(range(i, min(N, i + G))) for i in range(0, N, G)]
with ThreadPoolExecutor(max_workers=T) as pool:
    for k in range(sweeps):
        t0 = time.time_ns(); c0 = time.perf_counter()
        list(pool.map(one, groups))
        dt = time.perf_counter() - c0
        print(f"SWEEP {k} {t0} {time.time_ns()} {1e3*dt:.1f} ms", flush=True)
