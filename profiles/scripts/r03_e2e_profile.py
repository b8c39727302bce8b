import os, sys, time; sys.path.insert(0, os.getcwd())
import cProfile, pstats
from score_amd.manhattan import make_manhattan
from score_amd import solve_score as ss
from score_amd.native import ArrayGraph, graph_arrays
trials = [make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=5000 + t) for t in range(64)]
flat = [ArrayGraph(graph_arrays(fg)) for fg in trials]
st = dict(eps_abs=1e-7, eps_rel=1e-7)
ss.solve_score_batch(flat, "SOCP", solver_settings=st, workers=1)
pr = cProfile.Profile(); pr.enable()
ss.solve_score_batch(flat, "SOCP", solver_settings=st, workers=1)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
