"""PCG iterations used against queued, per Newton iteration (verbose log of one default solve)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from score_amd.manhattan import make_manhattan
from score_amd.native import assemble_native, graph_arrays
from score_amd.solver import ConicSolver
r = int(sys.argv[1]) if len(sys.argv) > 1 else 20
fg = make_manhattan(n_robots=r, n_poses=1000, n_beacons=4, seed=3000 if r == 20 else 5000)
m = assemble_native(fg, "SOCP", arrays=graph_arrays(fg))
s = ConicSolver([m.qp], dict(verbose=0)); s.solve(); s.close()
s = ConicSolver([m.qp], dict(verbose=1)); o = s.solve()[0]; s.close()
print(o.info["solve_ms"], o.info["newton_iters"], o.info["newton_cg_iters"])
