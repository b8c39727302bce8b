import sys, time; sys.path.insert(0,'.')
import numpy as np
from score_amd.manhattan import make_config, make_manhattan
from score_amd.assemble import assemble
from score_amd.solver import ConicSolver
from score_amd import io as sio
import os
graphs = {}
for c in (1,2,3): graphs['cfg%d'%c] = make_config(c)
for seed in range(3): graphs['mc%d'%seed] = make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=3000+seed)
from tests.conftest import graph_by_name
import glob
try:
    import tests.conftest as tc
    fx = tc.load_fixture_graphs() if hasattr(tc,'load_fixture_graphs') else None
except Exception as e:
    fx = None
for name, fg in graphs.items():
    qp = assemble(fg,'SOCP').qp
    row=[]
    for ci in (5, 10, 15, 25, 40):
        s = ConicSolver(qp, dict(check_interval=ci, adaptive_rho_interval=100)); s.solve()
        ts=[]
        for rep in range(3):
            out = s.solve()[0]; ts.append(out.info['solve_ms'])
        row.append('ci=%d: %.2f ms (admm %d, newton %d, cg %d, %s)'%(ci, min(ts), out.info['iters'], out.info['newton_iters'], out.info['newton_cg_iters'], 'ok' if out.solved else 'FAIL'))
        s.close()
    print(name, ' | '.join(row), flush=True)
