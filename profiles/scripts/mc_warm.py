import sys, time; sys.path.insert(0,'.')
from concurrent.futures import ThreadPoolExecutor
from score_amd.manhattan import make_manhattan
from score_amd.assemble import assemble
from score_amd.solver import ConicSolver
qps = [assemble(make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=4000+t),'SOCP').qp for t in range(64)]
for warm in (10, 15, 25, 40):
    solvers = [ConicSolver(qps[i:i+16], dict(polish_warmup=warm, check_interval=max(25,warm))) for i in range(0,64,16)]
    with ThreadPoolExecutor(4) as pool:
        sweep = lambda: [r for rs in pool.map(lambda s: s.solve(), solvers) for r in rs]
        sweep(); t=time.perf_counter(); 
        for _ in range(3): last = sweep()
        dt=time.perf_counter()-t
    print('polish_warmup %d: %.0f problems/s, solved %d, newton its per group %s'%(warm, 64*3/dt, sum(r.solved for r in last), [last[i].info['newton_iters'] for i in (0,16,32,48)]), flush=True)
    for s in solvers: s.close()
