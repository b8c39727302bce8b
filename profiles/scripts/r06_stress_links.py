"""Random graphs WITH loop closures -- 2-D and 3-D, 1-5 robots, 40-1500 poses, 1-14 loop closures (some beyond the caps of
csrc/score_link.hpp: those problems keep the chain preconditioner alone), lock-step batches of 1-3 -- through the default solver
with and without the link correction: every graph solved, objectives equal to 1e-6 relative.  python profiles/scripts/r06_stress_links.py"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
from score_amd.manhattan import make_manhattan, make_manhattan_3d
from score_amd.solve_score import solve_score_batch

rng = np.random.default_rng(606)
bad, t0 = [], time.time()
tot = [0.0, 0.0]; pcg = [0, 0]
for trial in range(80):
    three = trial % 4 == 3
    cnt = int(rng.integers(1, 4))
    graphs = []
    for _ in range(cnt):
        R = int(rng.integers(1, 6)); T = int(rng.integers(40, 900 if three else 1500)); Nb = int(rng.integers(1, 5)); nlc = int(rng.integers(1, 15))
        mk = make_manhattan_3d if three else make_manhattan
        graphs.append(mk(n_robots=R, n_poses=T, n_beacons=Nb, seed=int(rng.integers(0, 2**31)), p_range=float(rng.uniform(0.05, 0.4)), n_loop_closures=nlc))
    res = []
    try:
        for env in (None, "1"):
            if env: os.environ["SCORE_NO_LINKS"] = env
            else: os.environ.pop("SCORE_NO_LINKS", None)
            res.append(solve_score_batch(graphs, "SOCP", lockstep=True))
    except AssertionError as exc:  # (the reference's graph check: an unmeasured beacon)
        os.environ.pop("SCORE_NO_LINKS", None)
        print(trial, "skipped:", str(exc)[:60]); continue
    os.environ.pop("SCORE_NO_LINKS", None)
    for k, (a, b) in enumerate(zip(*res)):
        ok = a.solved and b.solved and abs(a.info["pobj"] - b.info["pobj"]) <= 1e-6 * max(1.0, abs(a.info["pobj"]))
        if not ok:
            bad.append((trial, k, a.info["status"], b.info["status"], a.info["pobj"], b.info["pobj"])); print("BAD", bad[-1], flush=True)
        pcg[0] += a.info["newton_cg_iters"]; pcg[1] += b.info["newton_cg_iters"]
    tot[0] += res[0][0].info["solve_ms"]; tot[1] += res[1][0].info["solve_ms"]
    if trial % 20 == 19: print(f"{trial + 1} batches done, {len(bad)} bad, {time.time() - t0:.0f} s", flush=True)
print(f"BAD: {bad}; Newton PCG iterations {pcg[0]} | {pcg[1]}, handle solve time {tot[0]:.0f} | {tot[1]:.0f} ms (with | without links)")
