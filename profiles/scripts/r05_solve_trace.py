"""Kernel-trace target: default solves of the headline problem (20 robots x 1000 poses), one handle.  python r05_solve_trace.py [reps]"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from score_amd.manhattan import make_manhattan
from score_amd.native import graph_arrays
from score_amd.solver import ConicSolver
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
s = ConicSolver.from_graphs([graph_arrays(make_manhattan(n_robots=20, n_poses=1000, n_beacons=4, seed=3000))], 0, {})
for _ in range(reps):
    r = s.solve()[0]
print("solve_ms", r.info["solve_ms"], "newton", r.info["newton_iters"], "pcg", r.info["newton_cg_iters"])
s.close()
