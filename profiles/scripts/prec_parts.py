import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from score_amd.manhattan import make_manhattan
from score_amd.assemble import assemble
from score_amd.solver import ConicSolver
for npose in (1000, 255):
    fg = make_manhattan(n_robots=20, n_poses=npose, n_beacons=4, seed=1)
    qp = assemble(fg,'SOCP').qp
    s = ConicSolver(qp, dict(max_iters=50, polish=0)); s.solve()
    for kind in ('prec_init', 'prec_step'):
        out = {m: round(s.debug_time(f"{kind}:{m}", 300)*1e3,2) for m in (0,1,2,3,4,7)}
        print(npose, kind, out, flush=True)
    s.close()
