import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..')))
LIB=os.environ['SCORE_TSAN_LIB']
from score_amd.manhattan import make_manhattan
from score_amd.native import assemble_native
from score_amd.solver import ConicSolver
from concurrent.futures import ThreadPoolExecutor
fgs = [make_manhattan(n_robots=4, n_poses=600, n_beacons=3, seed=i) for i in range(4)]
def one(fg):
    m = assemble_native(fg, "SOCP", lib_path=LIB)
    s = ConicSolver([m.qp], dict(max_iters=25), lib_path=LIB); s.solve(); s.close()
    return True
with ThreadPoolExecutor(4) as pool:
    print(list(pool.map(one, fgs)))
print(one(fgs[0]))
