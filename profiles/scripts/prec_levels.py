import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from score_amd.manhattan import make_manhattan
from score_amd.assemble import assemble
from score_amd.solver import ConicSolver
for npose in (1000, 255, 63, 15, 4):
    fg = make_manhattan(n_robots=20, n_poses=npose, n_beacons=4, seed=1)
    qp = assemble(fg,'SOCP').qp
    for radix in (4, 2):
        s = ConicSolver(qp, dict(max_iters=50, chain_radix=radix, polish=0)); s.solve()
        ti = s.debug_time("prec_init", 300)*1e3; ts = s.debug_time("prec_step", 300)*1e3
        print('poses/robot %4d radix %d: prec_init %.2f us prec_step %.2f us  n=%d'%(npose, radix, ti, ts, qp.n), flush=True)
        s.close()
