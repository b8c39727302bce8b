"""Where the end-to-end time of solve_score() goes (headline graph), on the GPU box's host."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cProfile, pstats
from score_amd.manhattan import make_manhattan
from score_amd import solve_score as ss
from score_amd.native import graph_arrays, assemble_native
fg = make_manhattan(n_robots=20, n_poses=1000, n_beacons=4, seed=3000)
ss.solve_score(fg, "SOCP")
for _ in range(2):
    t = time.perf_counter(); ss._check_factor_graph(fg); t1 = time.perf_counter() - t
    t = time.perf_counter(); arr = graph_arrays(fg); t2 = time.perf_counter() - t
    t = time.perf_counter(); m = assemble_native(fg, "SOCP", arrays=arr); t3 = time.perf_counter() - t
    print(f"check {t1*1e3:.1f} ms  graph_arrays {t2*1e3:.1f} ms  score_assemble + maps {t3*1e3:.1f} ms")
pr = cProfile.Profile(); pr.enable()
ss.solve_score(fg, "SOCP")
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
