import sys, time, ctypes as C
import os; sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from concurrent.futures import ThreadPoolExecutor
from score_amd.manhattan import make_manhattan
from score_amd.native import graph_arrays, score_graph_struct, _bind
from score_amd.solver import load_library
lib = load_library(None); _bind(lib)
fgs = [make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=5000+t) for t in range(8)]
arrs = [graph_arrays(fg) for fg in fgs]
gs = [score_graph_struct(a, 0) for a in arrs]
def one(i):
    t = time.perf_counter()
    h = C.c_void_p()
    assert lib.score_assemble(C.byref(gs[i % 8]), C.byref(h)) == 0
    dt = time.perf_counter() - t
    lib.score_assembled_free(h)
    return dt
for w in (1, 2, 4, 8, 16):
    for rep in range(3):
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=w) as pool: r = list(pool.map(one, range(64)))
        wall = time.perf_counter() - t0
    print(w, f"wall {1e3*wall:.1f} ms; avg call {1e3*sum(r)/64:.2f} ms")
