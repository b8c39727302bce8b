"""Larger than the headline: robots x 1000 poses, default solver -- solved?, iterations, time, setup (one MI355X)."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from score_amd.manhattan import make_manhattan
from score_amd.native import assemble_native, graph_arrays
from score_amd.solver import ConicSolver
for robots, poses in ((20, 1000), (60, 1000), (120, 1000), (20, 5000)):
    t = time.perf_counter(); fg = make_manhattan(n_robots=robots, n_poses=poses, n_beacons=4, seed=3000); tg = time.perf_counter() - t
    t = time.perf_counter(); m = assemble_native(fg, "SOCP", arrays=graph_arrays(fg)); ta = time.perf_counter() - t
    t = time.perf_counter(); s = ConicSolver([m.qp], {}); tc = time.perf_counter() - t
    s.solve()
    o = min((s.solve()[0] for _ in range(3)), key=lambda r: r.info["solve_ms"])
    s.close()
    a = ConicSolver([m.qp], dict(polish=0)); a.solve()
    t = time.perf_counter(); oa = a.solve()[0]; dta = time.perf_counter() - t
    a.close()
    print(f"{robots} robots x {poses} poses: n {m.qp.n}, m {m.qp.m}; assemble {1e3*ta:.0f} ms, create {1e3*tc:.0f} ms; default solve {o.info['solve_ms']:.1f} ms "
          f"(solved {o.solved}, {o.info['iters']} ADMM + {o.info['newton_iters']} Newton, {o.info['newton_cg_iters']} PCG); "
          f"ADMM alone {oa.info['iters']} iterations, {oa.info['iters']/dta:.0f} it/s, solved {oa.solved}", flush=True)
