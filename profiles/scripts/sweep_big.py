import sys, time; sys.path.insert(0,'.')
import numpy as np
from score_amd.manhattan import make_manhattan
from score_amd.assemble import assemble
from score_amd.solver import ConicSolver
rng = np.random.default_rng(77)
qps=[]; kws=[]
while len(qps) < 144:
    kw = dict(n_robots=int(rng.integers(1,6)), n_poses=int(rng.integers(5,400)), n_beacons=int(rng.integers(0,6)), seed=int(rng.integers(0,100000)),
              p_range=float(rng.choice([0.05,0.1,0.2,0.4,0.8])), n_loop_closures=int(rng.choice([0,0,0,2,5])))
    fg = make_manhattan(**kw)
    if fg.unconnected_variable_names: continue
    qps.append(assemble(fg,'SOCP').qp); kws.append(kw)
t0=time.time()
single=[]
for qp in qps:
    s=ConicSolver(qp, {}); single.append(s.solve()[0]); s.close()
t1=time.time()
batch=[]
for i in range(0,len(qps),12):
    s=ConicSolver(qps[i:i+12], {}); batch += s.solve(); s.close()
t2=time.time()
bad=0; worst=0
for i,(a,b) in enumerate(zip(single,batch)):
    rel=abs(a.info['pobj']-b.info['pobj'])/max(1.0,abs(a.info['pobj']))
    worst=max(worst,rel)
    if not (a.solved and b.solved and rel<1e-6):
        bad+=1; print('CHECK', i, kws[i], a.solved, b.solved, a.info['pobj'], b.info['pobj'], a.info['iters'], a.info['newton_iters'], b.info['iters'], b.info['newton_iters'])
print('graphs %d: single %.2f s, batch(12) %.2f s, bad %d, worst rel objective diff %.1e, newton iters single max %d, polish used in %d'%(len(qps), t1-t0, t2-t1, bad, worst, max(a.info['newton_iters'] for a in single), sum(a.info['newton_iters']>0 for a in single)))
# ADMM-only cross-check on a subset
cnt=0; w2=0
for i in range(0,len(qps),6):
    s=ConicSolver(qps[i], dict(polish=0, max_iters=20000)); o=s.solve()[0]; s.close()
    if o.solved:
        cnt+=1; w2=max(w2, abs(o.info['pobj']-single[i].info['pobj'])/max(1.0,abs(o.info['pobj'])))
print('ADMM-only converged on %d of %d sampled, worst objective diff vs polished %.1e'%(cnt, len(range(0,len(qps),6)), w2))
