"""144 random graphs (1-5 robots, 5-400 poses, 0-5 beacons, loop closures): the product default solver with the
chain factors kept to float precision (fac_fp32 = 1, default) against factors in double -- solved flags,
objectives, iteration counts, and the solver's own residuals.  (Certificates from the oracle: the -m gpu tests,
test_random_graphs_against_the_oracle.)"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import numpy as np
from score_amd.assemble import assemble
from score_amd.manhattan import make_manhattan
from score_amd.solver import ConicSolver
rng = np.random.default_rng(77)
qps, kws = [], []
while len(qps) < 144:
    kw = dict(n_robots=int(rng.integers(1, 6)), n_poses=int(rng.integers(5, 400)), n_beacons=int(rng.integers(0, 6)), seed=int(rng.integers(0, 100000)),
              p_range=float(rng.choice([0.05, 0.1, 0.2, 0.4, 0.8])), n_loop_closures=int(rng.choice([0, 0, 0, 2, 5])))
    fg = make_manhattan(**kw)
    if fg.unconnected_variable_names:
        continue
    qps.append(assemble(fg, "SOCP").qp); kws.append(kw)
res = {}
for fp32 in (1, 0):
    out = []
    t0 = time.time()
    for qp in qps:
        s = ConicSolver(qp, dict(fac_fp32=fp32)); out.append(s.solve()[0]); s.close()
    res[fp32] = out
    print(f"fac_fp32={fp32}: {time.time()-t0:.2f} s, solved {sum(o.solved for o in out)}/{len(out)}, ADMM iterations {sum(o.info['iters'] for o in out)}, "
          f"Newton iterations {sum(o.info['newton_iters'] for o in out)}, Newton PCG iterations {sum(o.info['newton_cg_iters'] for o in out)}", flush=True)
worst = max(abs(a.info["pobj"] - b.info["pobj"]) / max(1.0, abs(b.info["pobj"])) for a, b in zip(res[1], res[0]))
print(f"worst relative objective difference fp32 vs fp64 factors: {worst:.1e}")
wp = max(o.info["res_pri"] for o in res[1]); wd = max(o.info["res_dual"] for o in res[1])
print(f"default solves: worst primal residual {wp:.1e}, worst dual residual {wd:.1e} (solver's own, unscaled)")
# ADMM-only (polish off), a subset: iterations to eps with both factor precisions
for fp32 in (1, 0):
    tot = conv = 0
    for qp in qps[::6]:
        s = ConicSolver(qp, dict(polish=0, fac_fp32=fp32, max_iters=20000)); o = s.solve()[0]; s.close()
        tot += o.info["iters"]; conv += int(o.solved)
    print(f"ADMM only, fac_fp32={fp32}: converged {conv}/{len(qps[::6])}, iterations {tot}")
