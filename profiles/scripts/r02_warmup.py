"""Product default solve time against the number of ADMM warm-up iterations before the Newton polish."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from score_amd.assemble import assemble
from score_amd.manhattan import make_manhattan
from score_amd.solver import ConicSolver
for (r, n, b, seed) in ((20, 1000, 4, 3000), (4, 1000, 4, 4000), (4, 1000, 4, 4001), (4, 1000, 4, 4002)):
    qp = assemble(make_manhattan(n_robots=r, n_poses=n, n_beacons=b, seed=seed), "SOCP").qp
    for wu in (5, 10, 15, 20, 30):
        s = ConicSolver(qp, dict(polish_warmup=wu)); s.solve()
        t0 = time.perf_counter()
        for _ in range(8): o = s.solve()[0]
        dt = (time.perf_counter() - t0) / 8
        i = o.info
        print(f"{r}x{n} seed {seed} warmup {wu:2d}: {dt*1e3:.2f} ms solved={o.solved} admm={i['iters']} newton={i['newton_iters']} pcg={i['newton_cg_iters']}", flush=True)
        s.close()
