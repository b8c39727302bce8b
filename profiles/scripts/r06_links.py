"""Loop closures inside the Newton preconditioner (csrc/score_link.hpp), A/B against SCORE_NO_LINKS=1 on round 5's stress graphs
(profiles/scripts/r05_stress_3d_long.py: 36 random graphs, 2-D with 0-3 loop closures and chains of up to 3200 poses, 3-D without)
plus 2-D / 3-D graphs with more loop closures.  Columns: Newton iterations, Newton PCG iterations and solve_ms, with | without.
python profiles/scripts/r06_links.py"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
from score_amd.manhattan import make_manhattan, make_manhattan_3d
from score_amd.solve_score import solve_score

def both(fg):
    out = []
    for env in (None, "1"):
        if env: os.environ["SCORE_NO_LINKS"] = env
        else: os.environ.pop("SCORE_NO_LINKS", None)
        solve_score(fg, "SOCP")
        r = solve_score(fg, "SOCP")
        out.append(r)
    os.environ.pop("SCORE_NO_LINKS", None)
    return out

rng = np.random.default_rng(77)
tot = [0, 0]; ms = [0.0, 0.0]
for trial in range(36):
    three = trial % 3 == 0
    R = int(rng.integers(1, 5)); Nb = int(rng.integers(1, 5))
    T = int(rng.integers(30, 1500 if three else 3200))
    if three:
        fg = make_manhattan_3d(n_robots=R, n_poses=T, n_beacons=Nb, seed=1000 + trial, p_range=float(rng.uniform(0.05, 0.4))); nlc = 0
    else:
        nlc = None
        p = float(rng.uniform(0.05, 0.4)); nlc = int(rng.integers(0, 4))
        fg = make_manhattan(n_robots=R, n_poses=T, n_beacons=Nb, seed=1000 + trial, p_range=p, n_loop_closures=nlc)
    if nlc == 0:
        continue
    try:
        a, b = both(fg)
    except AssertionError as exc:
        print(trial, "skipped:", str(exc)[:60]); continue
    ok = a.solved and b.solved and abs(a.info["pobj"] - b.info["pobj"]) <= 1e-6 * max(1.0, abs(a.info["pobj"]))
    tot[0] += a.info["newton_cg_iters"]; tot[1] += b.info["newton_cg_iters"]; ms[0] += a.info["solve_ms"]; ms[1] += b.info["solve_ms"]
    print(f"{trial:2d} 2-D {R} x {T:4d}, {Nb} beacons, {nlc} loop closures: newton {a.info['newton_iters']:2d} | {b.info['newton_iters']:2d}  pcg {a.info['newton_cg_iters']:4d} | {b.info['newton_cg_iters']:4d}"
          f"  per newton {a.info['newton_cg_iters'] / max(1, a.info['newton_iters']):5.1f} | {b.info['newton_cg_iters'] / max(1, b.info['newton_iters']):5.1f}  ms {a.info['solve_ms']:6.2f} | {b.info['solve_ms']:6.2f}  {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"stress graphs with loop closures: pcg {tot[0]} | {tot[1]}, solve ms {ms[0]:.1f} | {ms[1]:.1f}")
for name, fg in (("2-D 4 x 800, 8 loop closures", make_manhattan(n_robots=4, n_poses=800, n_beacons=4, seed=9, n_loop_closures=8)),
                 ("2-D 2 x 400, 2 loop closures", make_manhattan(n_robots=2, n_poses=400, n_beacons=3, seed=10, n_loop_closures=2)),
                 ("2-D 4 x 1000, 20 loop closures (8 inside the cap)", make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=11, n_loop_closures=20)),
                 ("3-D 3 x 600, 4 loop closures", make_manhattan_3d(n_robots=3, n_poses=600, n_beacons=4, seed=12, n_loop_closures=4))):
    a, b = both(fg)
    print(f"{name}: newton {a.info['newton_iters']} | {b.info['newton_iters']}  pcg {a.info['newton_cg_iters']} | {b.info['newton_cg_iters']}  per newton "
          f"{a.info['newton_cg_iters'] / max(1, a.info['newton_iters']):.1f} | {b.info['newton_cg_iters'] / max(1, b.info['newton_iters']):.1f}  ms {a.info['solve_ms']:.2f} | {b.info['solve_ms']:.2f}  "
          f"pobj {a.info['pobj']:.9g} | {b.info['pobj']:.9g}", flush=True)
