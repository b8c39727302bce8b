"""Kernel-trace target: 400 ADMM iterations of 20 robots x 5000 poses (segmented chains + second level, csrc/score_join.hpp) or of
the 3-D leg (4 x 1000 poses); prints the in-loop dispatch times.  python r05_long_trace.py long|3d"""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
from score_amd.assemble import assemble
from score_amd.native import assemble_native
from score_amd.manhattan import make_manhattan, make_manhattan_3d
from score_amd.solver import ConicSolver
which = sys.argv[1]
if which == "long":
    m = assemble_native(make_manhattan(n_robots=20, n_poses=5000, n_beacons=4, seed=0), "SOCP")
else:
    m = assemble_native(make_manhattan_3d(n_robots=4, n_poses=1000, n_beacons=4, seed=7000), "SOCP")
s = ConicSolver([m.qp], dict(polish=0, adaptive_rho=0, adaptive_cg=0))
s.steps(400)
dev, disp = s.time_iteration(warmup=10, iters=40, dispatch=True)
print(which, "dispatch us:", {k: round(v, 2) for k, v in disp.items()}, "sum %.1f" % sum(disp.values()))
s.close()
