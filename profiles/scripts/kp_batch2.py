import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from score_amd.manhattan import make_manhattan
from score_amd.assemble import assemble
from score_amd.solver import ConicSolver
for B in (1, 4, 16):
    qps = [assemble(make_manhattan(n_robots=20, n_poses=1000, n_beacons=4, seed=3000+i),'SOCP').qp for i in range(B)]
    s = ConicSolver(qps, dict(polish=0, max_iters=100)); s.solve()
    ms, by = s.time_kkt_apply(300)
    us = s.time_iteration(20, 60)
    print('B=%2d: KP back-to-back %.2f us (%.0f GB/s, %.3f) | in loop (device clock) KP %.2f us (%.0f GB/s, %.3f) | in loop all: %s sum %.1f'%(B, ms*1e3, by/ms/1e6, by/ms/1e6/8000, us['kp'], by/us['kp']/1e3, by/us['kp']/1e3/8000, {k: round(v,1) for k,v in us.items()}, sum(us.values())), flush=True)
    s.close()
