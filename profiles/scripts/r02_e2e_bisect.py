"""Which earlier leg of bench.py slows the Python side of solve_score() afterwards."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import bench
args = bench.parse_args([])
D = bench.Dist(args)
from score_amd.manhattan import make_manhattan
from score_amd.solve_score import solve_score
from score_amd.solver import ConicSolver
fg = make_manhattan(n_robots=args.robots, n_poses=args.poses, n_beacons=args.beacons, seed=3000)
st = dict(device=0, eps_abs=args.eps, eps_rel=args.eps)
def e2e(tag):
    ts = []
    for _ in range(4):
        t = time.perf_counter(); r = solve_score(fg, "SOCP", solver_settings=st); ts.append(1e3 * (time.perf_counter() - t))
    print(f"{tag:36s} total {min(ts):6.1f}..{max(ts):6.1f} ms  create {r.info['setup_ms']:.1f} solve {r.info['solve_ms']:.1f}", flush=True)
solve_score(fg, "SOCP", solver_settings=st)
e2e("start")
models = bench.make_headline(args, 0, 1)
e2e("after make_headline")
base = bench.base_settings(args, 0) if hasattr(bench, "base_settings") else dict(device=0, eps_abs=args.eps, eps_rel=args.eps)
solver = ConicSolver([m.qp for m in models], dict(base, polish=0))
for _ in range(3): solver.solve()
e2e("after ADMM-only solves (handle live)")
solver.time_kkt_apply(args.kkt_reps); solver.time_iteration(warmup=50, iters=200, dispatch=True)
e2e("after probes")
solver.close()
e2e("after close")
bm = models + bench.make_headline(args, 1, 15)
bs_ = ConicSolver([m.qp for m in bm], dict(base, polish=0, adaptive_cg=0)); bs_.time_iteration(warmup=10, iters=40, dispatch=True); bs_.close()
e2e("after batch16 (models live)")
del bm
e2e("after del bm")
bench.montecarlo_on_this_gpu(args, 0)
e2e("after montecarlo")
import gc; gc.collect()
e2e("after gc.collect")
