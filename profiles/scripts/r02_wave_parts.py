"""Phase breakdown of the split chain kernel (debug_skip bits: 4 no solve phases, 32 no poll, 64 no exchange at all)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from score_amd.manhattan import make_manhattan
from score_amd.assemble import assemble
from score_amd.solver import ConicSolver
fg = make_manhattan(n_robots=20, n_poses=1000, n_beacons=4, seed=1)
qp = assemble(fg, 'SOCP').qp
for split in (1, 0):
    s = ConicSolver(qp, dict(max_iters=50, polish=0, chain_split=split)); s.solve()
    for kind in ('prec_init', 'prec_step'):
        out = {m: round(s.debug_time(f"{kind}:{m}", 300) * 1e3, 2) for m in ((0, 32, 64, 4) if split else (0, 4, 7))}
        print('split' if split else 'unsplit', kind, out, flush=True)
    s.close()
