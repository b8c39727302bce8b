import sys, os, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from score_amd.io import load_fg_npz
from score_amd.assemble import assemble
from score_amd.solver import ConicSolver
from score_amd.manhattan import make_manhattan
G='tests/golden'
graphs = {'manhattan fixture (4 x 400)': load_fg_npz(os.path.join(G,'manhattan_fg.npz')), 'GOATS': load_fg_npz(os.path.join(G,'goats_fg.npz')),
          'degenerate 3-robot graph (seed 302)': make_manhattan(n_robots=3, n_poses=60, n_beacons=4, seed=302, p_range=0.4)}
for name, fg in graphs.items():
    qp = assemble(fg,'SOCP').qp
    s = ConicSolver(qp, {}); s.solve(); o = s.solve()[0]; s.close()
    s = ConicSolver(qp, dict(polish=0)); s.solve(); a = s.solve()[0]; s.close()
    print('%s: n=%d | full solver %.2f ms (admm %d, newton %d, cg %d, solved %s) | ADMM alone %.1f ms (%d its, solved %s)'%(name, qp.n, o.info['solve_ms'], o.info['iters'], o.info['newton_iters'], o.info['newton_cg_iters'], o.solved, a.info['solve_ms'], a.info['iters'], a.solved), flush=True)
