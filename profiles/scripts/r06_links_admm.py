"""Loop closures inside the ADMM loop's preconditioner (the K set of csrc/score_link.hpp): the ADMM loop ALONE (polish off,
adaptive penalty and PCG count on: every penalty change refreshes the correction) with | without SCORE_NO_LINKS=1 on graphs
with loop closures.  Columns: ADMM iterations, PCG iterations, solve_ms.  python profiles/scripts/r06_links_admm.py"""
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
        st = dict(polish=0, max_iters=60000)
        solve_score(fg, "SOCP", solver_settings=st)
        out.append(solve_score(fg, "SOCP", solver_settings=st))
    os.environ.pop("SCORE_NO_LINKS", None)
    return out

rng = np.random.default_rng(78)
it = [0, 0]; pcg = [0, 0]; ms = [0.0, 0.0]; bad = 0
for trial in range(24):
    three = trial % 4 == 3
    R = int(rng.integers(1, 5)); Nb = int(rng.integers(1, 5)); T = int(rng.integers(40, 700 if three else 1500)); nlc = int(rng.integers(1, 9))
    mk = make_manhattan_3d if three else make_manhattan
    fg = mk(n_robots=R, n_poses=T, n_beacons=Nb, seed=2000 + trial, p_range=float(rng.uniform(0.05, 0.4)), n_loop_closures=nlc)
    try:
        a, b = both(fg)
    except AssertionError as exc:
        print(trial, "skipped:", str(exc)[:60]); continue
    ok = a.solved and b.solved and abs(a.info["pobj"] - b.info["pobj"]) <= 1e-5 * max(1.0, abs(a.info["pobj"]))
    bad += not ok
    it[0] += a.info["iters"]; it[1] += b.info["iters"]; pcg[0] += a.info["cg_iters"]; pcg[1] += b.info["cg_iters"]
    ms[0] += a.info["solve_ms"]; ms[1] += b.info["solve_ms"]
    print(f"{trial:2d} {'3-D' if three else '2-D'} {R} x {T:4d}, {Nb} beacons, {nlc} loop closures: admm {a.info['iters']:5d} | {b.info['iters']:5d}  pcg {a.info['cg_iters']:6d} | {b.info['cg_iters']:6d}"
          f"  ms {a.info['solve_ms']:7.2f} | {b.info['solve_ms']:7.2f}  status {a.info['status']} | {b.info['status']}  {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"ADMM alone on loop-closure graphs: iterations {it[0]} | {it[1]}, pcg {pcg[0]} | {pcg[1]}, solve ms {ms[0]:.1f} | {ms[1]:.1f}, mismatches {bad} (with | without links)")
