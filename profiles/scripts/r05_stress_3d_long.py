"""36 random graphs from the Python generators -- 2-D with loop closures and chains of up to 3200 poses, 3-D with chains of up to 1500 --
solved as SOCP, as direct QCQP (head form) and with SCORE_NO_SEGMENTS=1 (streaming chain kernel): the objectives must agree.
python profiles/scripts/r05_stress_3d_long.py   (columns: Newton iterations and solve_ms of the three runs)"""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
from score_amd.manhattan import make_manhattan, make_manhattan_3d
from score_amd.solve_score import solve_score, solve_score_batch
rng = np.random.default_rng(77)
bad = []
t0 = time.time()
for trial in range(36):
    three = trial % 3 == 0
    R = int(rng.integers(1, 5)); Nb = int(rng.integers(1, 5))
    T = int(rng.integers(30, 1500 if three else 3200))
    if three:
        fg = make_manhattan_3d(n_robots=R, n_poses=T, n_beacons=Nb, seed=1000 + trial, p_range=float(rng.uniform(0.05, 0.4)))
    else:
        fg = make_manhattan(n_robots=R, n_poses=T, n_beacons=Nb, seed=1000 + trial, p_range=float(rng.uniform(0.05, 0.4)), n_loop_closures=int(rng.integers(0, 4)))
    try:
        a = solve_score(fg, "SOCP")
        b = solve_score(fg, "QCQP", qcqp_mode="direct")
        os.environ["SCORE_NO_SEGMENTS"] = "1"
        c = solve_score(fg, "SOCP")
        os.environ.pop("SCORE_NO_SEGMENTS")
    except AssertionError as exc:  # (the reference's graph check: an unmeasured beacon)
        os.environ.pop("SCORE_NO_SEGMENTS", None)
        print(trial, "skipped:", str(exc)[:60], flush=True); continue
    ok = a.solved and b.solved and c.solved
    tol = 1e-6 * max(1.0, abs(a.info["pobj"]))
    if not ok or abs(a.info["pobj"] - b.info["pobj"]) > tol or abs(a.info["pobj"] - c.info["pobj"]) > tol:
        bad.append((trial, three, R, T, Nb, a.info["status"], b.info["status"], c.info["status"], a.info["pobj"], b.info["pobj"], c.info["pobj"]))
        print("BAD", bad[-1], flush=True)
    print(trial, "3-D" if three else "2-D", R, T, Nb, "newton", a.info["newton_iters"], b.info["newton_iters"], c.info["newton_iters"], "ms %.1f %.1f %.1f" % (a.info["solve_ms"], b.info["solve_ms"], c.info["solve_ms"]), flush=True)
print("BAD:", bad, "%.0f s" % (time.time() - t0))
