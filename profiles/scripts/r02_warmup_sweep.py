"""144 random graphs: product default solver with 10 / 15 / 20 warm-up iterations -- solved, iterations, solve time."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import numpy as np
from score_amd.assemble import assemble
from score_amd.manhattan import make_manhattan
from score_amd.solver import ConicSolver
rng = np.random.default_rng(77)
qps = []
while len(qps) < 144:
    kw = dict(n_robots=int(rng.integers(1, 6)), n_poses=int(rng.integers(5, 400)), n_beacons=int(rng.integers(0, 6)), seed=int(rng.integers(0, 100000)),
              p_range=float(rng.choice([0.05, 0.1, 0.2, 0.4, 0.8])), n_loop_closures=int(rng.choice([0, 0, 0, 2, 5])))
    fg = make_manhattan(**kw)
    if fg.unconnected_variable_names:
        continue
    qps.append(assemble(fg, "SOCP").qp)
for rep in range(2):
    for wu in (4, 6, 8, 10, 15):
        tot = 0.0; out = []
        for qp in qps:
            s = ConicSolver(qp, dict(polish_warmup=wu)); o = s.solve()[0]; s.close()
            out.append(o); tot += o.info["solve_ms"]
        print(f"warmup {wu}: solve time {tot:.1f} ms total, solved {sum(o.solved for o in out)}/144, ADMM {sum(o.info['iters'] for o in out)}, "
              f"Newton {sum(o.info['newton_iters'] for o in out)}, PCG {sum(o.info['newton_cg_iters'] for o in out)}", flush=True)
