"""How fast would the chain kernels be if every 1000-pose chain were four 250-pose chains?  Same number of
poses (20 000 x 2 rows), 40 chains of 1000 vs 160 chains of 250 vs 320 of 125 (all <= 256 + workgroups resident)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from score_amd.manhattan import make_manhattan
from score_amd.assemble import assemble
from score_amd.solver import ConicSolver
for robots, npose in ((20, 1000), (40, 500), (80, 250), (120, 167), (160, 125)):
    fg = make_manhattan(n_robots=robots, n_poses=npose, n_beacons=4, seed=1, p_range=0.1 * 20 / robots)
    qp = assemble(fg, 'SOCP').qp
    s = ConicSolver(qp, dict(max_iters=50, polish=0)); s.solve()
    ti = s.debug_time("prec_init", 300) * 1e3; ts = s.debug_time("prec_step", 300) * 1e3
    print(f"{robots:4d} robots x {npose:5d} poses: prec_init {ti:6.2f} us  prec_step {ts:6.2f} us   n={qp.n} chains={len(qp.chain_ptr)-1}", flush=True)
    s.close()
