import sys, time; sys.path.insert(0,'.')
from score_amd.manhattan import make_manhattan
from score_amd.assemble import assemble
from score_amd.solver import ConicSolver
qps = [assemble(make_manhattan(n_robots=20, n_poses=1000, n_beacons=4, seed=3000+i),'SOCP').qp for i in range(8)]
for B in (1, 2, 4, 8):
    s = ConicSolver(qps[:B], {}); s.solve(); o = s.solve()
    print('B=%d headline-size problems, lock-step full solver: %.2f ms -> %.1f problems/s, solved %s, newton %d cg %d'%(B, o[0].info['solve_ms'], 1e3*B/o[0].info['solve_ms'], all(x.solved for x in o), o[0].info['newton_iters'], o[0].info['newton_cg_iters']), flush=True)
    s.close()
