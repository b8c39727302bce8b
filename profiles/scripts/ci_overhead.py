import sys; sys.path.insert(0,'.')
from score_amd.manhattan import make_config
from score_amd.assemble import assemble
from score_amd.solver import ConicSolver
qp = assemble(make_config(3),'SOCP').qp
for ci in (25, 50, 100):
    s = ConicSolver(qp, dict(polish=0, check_interval=ci, adaptive_rho_interval=100)); s.solve(); o=s.solve()[0]
    print('check_interval %d: %d iterations %.2f ms -> %.2f us/it (%.0f it/s) solved %s'%(ci, o.info['iters'], o.info['solve_ms'], 1e3*o.info['solve_ms']/o.info['iters'], o.info['iters']/o.info['solve_ms']*1e3, o.solved), flush=True)
    s.close()
