"""Shorter segments than the LDS limit asks for (SCORE_SEG_NODES): the chain kernel on segments of <= S nodes + the join level,
against whole chains, on the headline problem, a 3-D problem and 64 config-5 trials.  One process per setting (the value is
read once).  python r05_seg_nodes.py <S>"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
S = sys.argv[1]
os.environ["SCORE_SEG_NODES"] = S
import numpy as np
from score_amd.native import assemble_native
from score_amd.manhattan import make_manhattan, make_manhattan_3d
from score_amd.solver import ConicSolver


def run(tag, qps, st_loop, n_loop):
    p = ConicSolver(qps, dict(st_loop, polish=0, max_iters=n_loop, eps_abs=1e-30, eps_rel=1e-30, check_interval=n_loop))
    p.solve()
    t = []
    for _ in range(3):
        t0 = time.perf_counter(); p.solve(); t.append(time.perf_counter() - t0)
    p.close()
    d = ConicSolver(qps, {})
    d.solve()
    ms, nw, ok = [], 0, True
    for _ in range(5):
        r = d.solve()
        ms.append(max(x.info["solve_ms"] for x in r)); nw = max(x.info["newton_iters"] for x in r); ok = ok and all(x.solved for x in r)
    d.close()
    print(f"S={S:>5} {tag:<10} loop {n_loop / min(t):9.0f} it/s   default solve {np.median(ms):6.2f} ms  newton {nw} solved {ok}", flush=True)


run("headline", [assemble_native(make_manhattan(n_robots=20, n_poses=1000, n_beacons=4, seed=0), "SOCP").qp], {}, 400)
run("3-D", [assemble_native(make_manhattan_3d(n_robots=4, n_poses=1000, n_beacons=4, seed=0), "SOCP").qp], {}, 400)
run("cfg5 x16", [assemble_native(make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=5000 + k), "SOCP").qp for k in range(16)], {}, 200)
