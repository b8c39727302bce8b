"""Loop-closure graphs: what the ADMM warm-up's PCG count costs the default solver now that the Newton preconditioner sees the
loop closures.  solve_score starts such graphs with cg_iters = 16 (a setting from the time the ADMM loop had to cope alone);
here the same graphs with cg_iters 2 / 4 / 8 / 16: Newton iterations, Newton PCG iterations, solve_ms.
python profiles/scripts/r06_lc_warmup.py"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
from score_amd.manhattan import make_manhattan, make_manhattan_3d
from score_amd.solve_score import solve_score

rng = np.random.default_rng(77)
graphs = []
for trial in range(36):
    three = trial % 3 == 0
    R = int(rng.integers(1, 5)); Nb = int(rng.integers(1, 5)); T = int(rng.integers(30, 1500 if three else 3200))
    if three:
        rng.uniform(0.05, 0.4); continue
    p = float(rng.uniform(0.05, 0.4)); nlc = int(rng.integers(0, 4))
    if nlc:
        graphs.append((f"{trial} 2-D {R} x {T}, {nlc} lc", make_manhattan(n_robots=R, n_poses=T, n_beacons=Nb, seed=1000 + trial, p_range=p, n_loop_closures=nlc)))
graphs.append(("3-D 3 x 600, 4 lc", make_manhattan_3d(n_robots=3, n_poses=600, n_beacons=4, seed=12, n_loop_closures=4)))
tot = {}
for name, fg in graphs:
    row = []
    for cg in (2, 4, 8, 16):
        st = dict(cg_iters=cg, cg_target=0.1)
        try:
            solve_score(fg, "SOCP", solver_settings=st)
            r = solve_score(fg, "SOCP", solver_settings=st)
        except AssertionError:
            row = None; break
        row.append((r.solved, r.info["newton_iters"], r.info["newton_cg_iters"], r.info["solve_ms"], r.info["pobj"]))
        tot[cg] = tot.get(cg, 0.0) + r.info["solve_ms"]
    if row:
        print(name, " | ".join(f"cg {cg}: {'ok' if s else 'NO'} n {n} pcg {p} {ms:.2f} ms" for cg, (s, n, p, ms, _) in zip((2, 4, 8, 16), row)),
              "obj spread %.1e" % (max(x[4] for x in row) - min(x[4] for x in row)), flush=True)
print("sum of solve_ms:", {k: round(v, 1) for k, v in tot.items()})
