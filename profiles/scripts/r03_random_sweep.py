"""Robustness sweep at the round's defaults (warm-up 6, re-factoring by active set, update helpers, XCD-aware tiles):
random 2-D and 3-D graphs, with and without loop closures / beacons; the default solver against the ADMM loop alone
(polish off) on the same problem: both solved, objectives equal to 1e-6 relative."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from score_amd.assemble import assemble
from score_amd.manhattan import make_manhattan, make_manhattan_3d
from score_amd.solver import ConicSolver
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(20260)
bad, worst, n_done, newton, t0 = [], 0.0, 0, 0, time.time()
while n_done < N:
    three_d = rng.random() < 0.25
    kw = dict(n_robots=int(rng.integers(1, 6)), n_poses=int(rng.integers(5, 300)), n_beacons=int(rng.integers(0, 5)), seed=int(rng.integers(0, 10**6)),
              p_range=float(rng.choice([0.05, 0.1, 0.3, 0.7])))
    if not three_d: kw["n_loop_closures"] = int(rng.choice([0, 0, 0, 2, 6]))
    fg = (make_manhattan_3d if three_d else make_manhattan)(**kw)
    if fg.unconnected_variable_names: continue
    qp = assemble(fg, "SOCP").qp
    extra = dict(cg_iters=16, cg_target=0.1) if kw.get("n_loop_closures") else {}
    a = ConicSolver(qp, dict(extra)); ra = a.solve()[0]; a.close()
    b = ConicSolver(qp, dict(extra, polish=0, max_iters=60000)); rb = b.solve()[0]; b.close()
    n_done += 1; newton += ra.info["newton_iters"]
    rel = abs(ra.info["pobj"] - rb.info["pobj"]) / max(1.0, abs(rb.info["pobj"]))
    if rb.solved: worst = max(worst, rel)
    if not ra.solved or (rb.solved and rel > 1e-6): bad.append((three_d, kw, ra.info["status"], rb.info["status"], rel))
print(f"{n_done} graphs in {time.time()-t0:.0f} s: default solver failures / objective mismatches {len(bad)}, worst relative objective difference {worst:.2e}, Newton iterations {newton}")
for x in bad[:10]: print("  ", x)
