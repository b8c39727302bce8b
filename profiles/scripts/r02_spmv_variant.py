"""In-loop kernel times (dispatch) of the ADMM iteration, single problem and a lock-step batch of 16, for a
given build of the library: python profiles/scripts/r02_spmv_variant.py [path/to/libscore_hip.so]"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from score_amd.assemble import assemble
from score_amd.manhattan import make_manhattan
from score_amd.solver import ConicSolver
lib = sys.argv[1] if len(sys.argv) > 1 else None
qps = [assemble(make_manhattan(n_robots=20, n_poses=1000, n_beacons=4, seed=3000 + i), "SOCP").qp for i in range(16)]
for B in (1, 16):
    s = ConicSolver(qps[:B], dict(polish=0, adaptive_cg=0), lib_path=lib)
    for rep in range(2):
        dev, disp = s.time_iteration(warmup=20, iters=100 if B == 1 else 40, dispatch=True)
    print(f"B={B:2d} dispatch us:", {k: round(v, 2) for k, v in disp.items()}, " sum", round(sum(disp.values()), 1), flush=True)
    s.close()
