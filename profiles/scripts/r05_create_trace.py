"""Kernel trace target: N creations of a handle from factor graphs (no solve).  python r05_create_trace.py [robots [batch [reps]]]"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from score_amd.manhattan import make_manhattan
from score_amd.native import graph_arrays
from score_amd.solver import ConicSolver

robots = int(sys.argv[1]) if len(sys.argv) > 1 else 20
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
arrs = [graph_arrays(make_manhattan(n_robots=robots, n_poses=1000, n_beacons=4, seed=3000 + t)) for t in range(batch)]
for _ in range(reps):
    ConicSolver.from_graphs(arrs, 0, {}).close()
