"""One graph with loop closures through the default solver, five times (profiler workload: the link kernels of csrc/score_link.hpp
beside the chain kernel).  2 robots x 2570 poses, 2 beacons, 3 loop closures (stress trial 2 of r05_stress_3d_long.py)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from score_amd.manhattan import make_manhattan
from score_amd.native import assemble_native
from score_amd.solver import ConicSolver
fg = make_manhattan(n_robots=2, n_poses=2570, n_beacons=2, seed=1002, p_range=0.08178057399708796, n_loop_closures=3)
s = ConicSolver([assemble_native(fg, "SOCP").qp], {})
for _ in range(5):
    o = s.solve()[0]
    print("solved", o.solved, "newton", o.info["newton_iters"], "pcg", o.info["newton_cg_iters"], "ms %.2f" % o.info["solve_ms"])
print("links", s.debug_get("links"))
s.close()
