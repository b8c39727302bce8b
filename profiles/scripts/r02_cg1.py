"""ADMM alone on the headline problem: PCG iterations per ADMM iteration vs iterations / time to eps."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from score_amd.manhattan import make_manhattan
from score_amd.native import assemble_native
from score_amd.solver import ConicSolver
for (r, n, seed) in ((20, 1000, 3000), (4, 1000, 4000)):
    qp = assemble_native(make_manhattan(n_robots=r, n_poses=n, n_beacons=4, seed=seed), "SOCP").qp
    for cg, adapt in ((2, 1), (1, 0), (1, 1), (3, 0)):
        s = ConicSolver(qp, dict(polish=0, cg_iters=cg, adaptive_cg=adapt)); s.solve()
        t0 = time.perf_counter(); o = s.solve()[0]; dt = time.perf_counter() - t0
        print(f"{r}x{n} cg_iters={cg} adaptive={adapt}: solved={o.solved} iters={o.info['iters']} pcg={o.info['cg_iters']} {dt*1e3:.1f} ms  {o.info['iters']/dt:.0f} it/s pobj={o.info['pobj']:.6f}", flush=True)
        s.close()
