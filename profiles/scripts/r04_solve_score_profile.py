"""cProfile of solve_score() on the headline graph (where the host side of one call goes).  python profiles/scripts/r04_solve_score_profile.py"""
import cProfile, os, pstats, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from score_amd.manhattan import make_manhattan
from score_amd.solve_score import solve_score
fg = make_manhattan(n_robots=20, n_poses=1000, n_beacons=4, seed=3000)
st = dict(device=0)
for _ in range(3): solve_score(fg, "SOCP", solver_settings=st)
ts = []
for _ in range(10):
    t = time.perf_counter(); solve_score(fg, "SOCP", solver_settings=st); ts.append(1e3 * (time.perf_counter() - t))
print("solve_score: min %.1f median %.1f ms" % (min(ts), sorted(ts)[5]))
pr = cProfile.Profile(); pr.enable()
for _ in range(5): solve_score(fg, "SOCP", solver_settings=st)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
