"""Consecutive default solves of 20 robots x 5000 poses with the solver's timeline on stderr (verbose): the reset line shows what
the first kernel of a solve waited for -- the stalls of profiles/r05_pageable_stalls.txt.  python r05_long_solves.py"""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
from score_amd.native import assemble_native
from score_amd.manhattan import make_manhattan
from score_amd.solver import ConicSolver
m = assemble_native(make_manhattan(n_robots=20, n_poses=5000, n_beacons=4, seed=0), "SOCP")
p = ConicSolver([m.qp], dict(verbose=1))
for k in range(4):
    print("=== solve", k, file=sys.stderr, flush=True)
    po = p.solve()[0]
    print("solve_ms %.2f" % po.info["solve_ms"], file=sys.stderr, flush=True)
p.close()
