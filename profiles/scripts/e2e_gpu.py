import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from score_amd.manhattan import make_config
from score_amd import solve_score as ss
from score_amd.assemble import assemble
from score_amd.solver import ConicSolver
fg = make_config(3)
for rep in range(6):
    t=time.time(); mdl = assemble(fg,'SOCP'); t_as=time.time()-t
    t=time.time(); sol = ConicSolver([mdl.qp], dict(verbose=(rep>=2))); t_cr=time.time()-t
    sol_settings_verbose_off = None
    t=time.time(); out = sol.solve()[0]; t_so=time.time()-t
    t=time.time(); res = ss.extract_solver_results(mdl, out.x, fg, 0.0, True, 'SOCP', out.info); t_ex=time.time()-t
    t=time.time(); sol.close(); t_cl=time.time()-t
    print('assemble %.3f create %.3f (setup_ms %.1f) solve %.4f (solve_ms %.2f) extract %.3f close %.3f'%(t_as,t_cr,out.info['setup_ms'],t_so,out.info['solve_ms'],t_ex,t_cl), flush=True)
for rep in range(3):
    t=time.time(); r = ss.solve_score(fg, 'SOCP'); print('solve_score e2e %.3f solved %s'%(time.time()-t, r.solved), flush=True)
for rep in range(3):
    t=time.time(); r = ss.solve_score(fg); print('solve_score (default QCQP via SOCP) e2e %.3f solved %s'%(time.time()-t, r.solved), flush=True)
