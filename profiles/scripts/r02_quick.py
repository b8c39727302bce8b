"""Quick timing of the product default solver (ADMM warm-up + Newton polish) on the BASELINE sizes,
and of BASELINE configs[4] on one GPU.  python profiles/scripts/r02_quick.py [mc]"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from score_amd.assemble import assemble
from score_amd.manhattan import make_manhattan
from score_amd.solver import ConicSolver

for (r, n, b, seed) in ((20, 1000, 4, 3000), (4, 1000, 4, 4000), (1, 500, 2, 1000)):
    qp = assemble(make_manhattan(n_robots=r, n_poses=n, n_beacons=b, seed=seed), "SOCP").qp
    s = ConicSolver(qp, {})
    s.solve()
    t0 = time.perf_counter()
    for _ in range(10):
        o = s.solve()[0]
    dt = (time.perf_counter() - t0) / 10
    i = o.info
    print(f"{r}x{n}: {dt*1e3:.2f} ms  solved={o.solved} admm={i['iters']} newton={i['newton_iters']} pcg={i['newton_cg_iters']} "
          f"pobj={i['pobj']:.9f} rp={i['res_pri']:.2e} rd={i['res_dual']:.2e}", flush=True)
    s.close()
if len(sys.argv) > 1:
    import bench
    from concurrent.futures import ThreadPoolExecutor
    for trials, per, thr in ((64, 16, 4), (32, 16, 4), (16, 16, 4), (8, 16, 4), (8, 8, 1), (8, 4, 2)):
        args = bench.parse_args(["--mc-batch", str(per), "--mc-threads", str(thr)])
        mc = bench.MonteCarlo(args, range(trials), 0, None)
        with ThreadPoolExecutor(max_workers=mc.threads) as pool:
            mc.sweep(pool)
            t0 = time.perf_counter()
            for _ in range(3):
                last = mc.sweep(pool)
            dt = time.perf_counter() - t0
        print(f"montecarlo {trials} trials groups {mc.group_sizes} threads {thr}: {trials*3/dt:.0f} problems/s, "
              f"{1e3*dt/3:.2f} ms per sweep, solved {sum(o.solved for o in last)}", flush=True)
        mc.close()
