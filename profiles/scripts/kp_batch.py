import sys, time; sys.path.insert(0,'.')
from score_amd.manhattan import make_manhattan
from score_amd.assemble import assemble
from score_amd.solver import ConicSolver
fg = [make_manhattan(n_robots=20, n_poses=1000, n_beacons=4, seed=3000+i) for i in range(16)]
qps = [assemble(f,'SOCP').qp for f in fg]
for B in (1,2,4,8,16):
    s = ConicSolver(qps[:B], dict(max_iters=25)); s.solve()
    ms, by = s.time_kkt_apply(300)
    print('batch', B, 'us %.2f'%(ms*1e3), 'GB/s %.0f'%(by/ms/1e6), 'frac %.3f'%(by/ms/1e6/8000), {k: round(s.debug_time(k,100)*1e3,1) for k in ('rhs','prec_init','prec_step','kpb','xupdate','cone')}, flush=True)
    s.close()
