import os, sys; sys.path.insert(0, os.getcwd())
from score_amd.manhattan import make_manhattan
from score_amd.native import assemble_native, graph_arrays
from score_amd.solver import ConicSolver
fgs = [make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=3000 + t) for t in range(16)]
ms = [assemble_native(fg, "SOCP", arrays=graph_arrays(fg)) for fg in fgs]
ConicSolver([m.qp for m in ms], {}).close()
ConicSolver([m.qp for m in ms], {}).close()
s = ConicSolver([m.qp for m in ms], dict(verbose=1)); s.close()
