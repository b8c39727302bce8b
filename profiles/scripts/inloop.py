import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from score_amd.manhattan import make_config
from score_amd.assemble import assemble
from score_amd.solver import ConicSolver
qp = assemble(make_config(3),'SOCP').qp
s = ConicSolver(qp, dict(polish=0)); o = s.solve()[0]
o = s.solve()[0]
print('solve_ms %.2f iters %d -> %.2f us/it'%(o.info['solve_ms'], o.info['iters'], 1e3*o.info['solve_ms']/o.info['iters']))
us = s.time_iteration(50, 200); print({k: round(v,2) for k,v in us.items()}, 'sum %.1f'%sum(us.values()))
s.close()
