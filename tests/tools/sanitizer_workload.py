import sys, numpy as np
ROOT=__import__('os').path.abspath(__import__('os').path.join(__import__('os').path.dirname(__file__), '..', '..')); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import os
LIB=os.environ['SCORE_ASAN_LIB']
from score_amd.manhattan import make_manhattan
from score_amd.solve_score import solve_score, solve_score_batch
from score_amd.refine import refine_estimate
from test_refine import _noisy_truth, check_refinement_edge_cases
from conftest import graph_3d
fg = make_manhattan(n_robots=3, n_poses=60, n_beacons=3, seed=5, p_range=0.4, n_loop_closures=3)
r = solve_score(fg, "SOCP", lib_path=LIB); print('solve', r.solved, r.info['iters'])
r = solve_score(fg, "QCQP", lib_path=LIB); print('qcqp', r.solved)
rs = solve_score_batch([make_manhattan(n_robots=2, n_poses=20+i, n_beacons=2, seed=i) for i in range(5)], "SOCP", lib_path=LIB, workers=2); print('batch', [x.solved for x in rs])
r3 = solve_score(graph_3d(n=20), "SOCP", lib_path=LIB); print('3d', r3.solved)
out, info = refine_estimate(fg, _noisy_truth(fg), lib_path=LIB); print('refine', info['iterations'], info['cost_final'])
check_refinement_edge_cases(LIB); print('edge ok')
from score_amd.rounding import round_to_special_orthogonal
from score_amd.solver import load_library
lib = load_library(LIB)
print('round', round_to_special_orthogonal(np.random.default_rng(0).normal(size=(100,3,3)), lib=lib).shape)
