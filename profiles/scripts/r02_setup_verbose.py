"""Host-side phases of model construction (score_assemble) and score_create on the headline graph."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
os.environ["SCORE_ASSEMBLE_VERBOSE"] = "1"
from score_amd.manhattan import make_manhattan
from score_amd.native import assemble_native, graph_arrays
from score_amd.solver import ConicSolver
fg = make_manhattan(n_robots=20, n_poses=1000, n_beacons=4, seed=3000)
arr = graph_arrays(fg)
for i in range(3):
    print(f"--- pass {i}", file=sys.stderr, flush=True)
    t = time.perf_counter(); m = assemble_native(fg, "SOCP", arrays=arr); t1 = time.perf_counter() - t
    t = time.perf_counter(); s = ConicSolver([m.qp], dict(verbose=1 if i == 2 else 0)); t2 = time.perf_counter() - t
    o = s.solve()[0]; s.close()
    print(f"assemble_native {t1*1e3:.1f} ms  create {t2*1e3:.1f} ms  solve {o.info['solve_ms']:.2f} ms", file=sys.stderr, flush=True)
