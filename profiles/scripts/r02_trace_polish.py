"""A few product-default solves of the headline problem, for a rocprofv3 --kernel-trace timeline."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from score_amd.assemble import assemble
from score_amd.manhattan import make_manhattan
from score_amd.solver import ConicSolver
r, n, b, seed = (20, 1000, 4, 3000) if len(sys.argv) < 2 else (4, 1000, 4, 4000)
qp = assemble(make_manhattan(n_robots=r, n_poses=n, n_beacons=b, seed=seed), "SOCP").qp
s = ConicSolver(qp, {})
for _ in range(4):
    t0 = time.perf_counter(); o = s.solve()[0]; print("solve ms", 1e3 * (time.perf_counter() - t0), o.info["newton_iters"], o.info["newton_cg_iters"])
s.close()
