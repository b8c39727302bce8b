"""Round-1 prototype kept as the evidence for "Anderson acceleration does not pay on SCORE instances" (DESIGN.md section 0):
type-II Anderson acceleration, safeguarded, on the fixed-point map of the direct-KKT ADMM of the same splitting (uses the oracle:
test / experiment infrastructure, not product).    python profiles/scripts/r04_anderson_proto.py"""
import sys, time, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from r04_admm_proto import ruiz, proj_soc_batch
from score_amd.assemble import assemble
from score_amd.manhattan import make_manhattan
from score_amd.io import load_fg_npz
from oracle import score_oracle as so

class Admm:
    def __init__(self, qp, rho=0.1, sigma=1e-6, alpha=1.6):
        self.qp=qp; P,A=qp.P,qp.A; self.n,self.m=qp.n,qp.m; self.dim=int(qp.soc_dims[0]) if len(qp.soc_dims) else 1
        self.D,self.E,self.Ps,self.As = ruiz(P,A,self.dim); self.qs=self.D*qp.q; self.bs=self.E*qp.b
        self.rho=rho; self.sigma=sigma; self.alpha=alpha
        self.lu = spla.splu((self.Ps + sigma*sp.identity(self.n) + rho*(self.As.T@self.As)).tocsc())
    def F(self, xi):
        n=self.n; x=xi[:n]; w=xi[n:]
        s = proj_soc_batch(w.reshape(-1,self.dim)).ravel(); y = self.rho*(s - w)
        rhs = self.sigma*x - self.qs + self.As.T@(self.rho*(self.bs - s) - y)
        xt = self.lu.solve(rhs)
        xn = self.alpha*xt + (1-self.alpha)*x
        v = self.alpha*(self.bs - self.As@xt) + (1-self.alpha)*s
        wn = v - y/self.rho
        return np.concatenate([xn, wn])
    def unscale(self, xi):
        n=self.n; x=xi[:n]; w=xi[n:]; s = proj_soc_batch(w.reshape(-1,self.dim)).ravel(); y=self.rho*(s-w)
        return self.D*x, s/self.E, self.E*y
    def residuals(self, xi):
        x,s,y=self.unscale(xi); qp=self.qp
        return np.max(np.abs(qp.A@x+s-qp.b)), np.max(np.abs(qp.P@x+qp.q+qp.A.T@y))

def run_plain(adm, iters, xref, npl):
    xi=np.zeros(adm.n+adm.m); hist=[]
    for k in range(1,iters+1):
        xi=adm.F(xi)
        if k%25==0:
            x,_,_=adm.unscale(xi); err=np.max(np.abs(x[:npl]-xref[:npl]))/np.max(np.abs(xref[:npl])); rp,rd=adm.residuals(xi); hist.append((k,rp,rd,err))
            if rp<1e-7 and rd<1e-6: break
    return hist

def run_aa(adm, iters, xref, npl, mem=10, every=1, reg=1e-10, safeguard=True):
    N=adm.n+adm.m; xi=np.zeros(N); hist=[]
    Xs=[]; Gs=[]  # history of xi and g = F(xi)-xi
    naa=0; nrej=0
    gnorm_prev=None
    for k in range(1,iters+1):
        Fx=adm.F(xi); g=Fx-xi
        Xs.append(xi.copy()); Gs.append(g.copy())
        if len(Xs)>mem+1: Xs.pop(0); Gs.pop(0)
        xi_new=Fx
        if len(Xs)>=3 and k%every==0:
            dG=np.stack([Gs[i+1]-Gs[i] for i in range(len(Gs)-1)],axis=1)
            dX=np.stack([Xs[i+1]-Xs[i] for i in range(len(Xs)-1)],axis=1)
            M=dG.T@dG; M+=reg*np.trace(M)/M.shape[0]*np.eye(M.shape[0])
            try:
                gam=np.linalg.solve(M, dG.T@g)
                cand = Fx - (dX+dG)@gam
                if safeguard:
                    gc = adm.F(cand)-cand
                    if np.linalg.norm(gc) <= 2.0*np.linalg.norm(g):
                        xi_new=cand; naa+=1
                    else:
                        nrej+=1; Xs=[]; Gs=[]
                else:
                    xi_new=cand; naa+=1
            except np.linalg.LinAlgError:
                pass
        xi=xi_new
        if k%25==0:
            x,_,_=adm.unscale(xi); err=np.max(np.abs(x[:npl]-xref[:npl]))/np.max(np.abs(xref[:npl])); rp,rd=adm.residuals(xi); hist.append((k,rp,rd,err))
            if rp<1e-7 and rd<1e-6: break
    return hist, naa, nrej

if __name__=='__main__':
    cases={}
    for seed in (302,306,311): cases[f's{seed}']=make_manhattan(n_robots=3, n_poses=40 + 10*(seed%4), n_beacons=4, seed=seed, p_range=0.4)
    cases['manh']=load_fg_npz(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests', 'golden') + '/manhattan_fg.npz')
    cases['goats']=load_fg_npz(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests', 'golden') + '/goats_fg.npz')
    for name,fg in cases.items():
        mdl=assemble(fg,'SOCP'); rp_,u,info=so.newton_solve(fg,tol=1e-13)
        vals=so.reduced_to_values(rp_,u,'SOCP'); xm=np.zeros(mdl.n_model)
        for i,nm in enumerate(mdl.pose_names): xm[i*6:(i+1)*6]=vals['poses'][nm].ravel()
        for i,nm in enumerate(mdl.landmark_names): xm[mdl.lm_base+i*2:mdl.lm_base+i*2+2]=vals['landmarks'][nm]
        for i,k in enumerate(mdl.range_keys): xm[mdl.rng_base+i]=vals['dists'][k][0]
        xref=mdl.reduce(xm); npl=len(mdl.pose_names)*6-6
        adm=Admm(mdl.qp, rho=0.03 if name.startswith('s') or name=='goats' else 0.1)
        h=run_plain(adm, 6000, xref, npl); print(name,'plain: iters',h[-1][0],'rp %.1e rd %.1e err %.1e'%h[-1][1:])
        for mem,every in ((5,1),(10,1),(10,10),(20,1)):
            h,naa,nrej=run_aa(adm, 6000, xref, npl, mem=mem, every=every); print(name,f'AA mem {mem} every {every}: iters',h[-1][0],'(F evals ~%d)'%(h[-1][0]+naa+nrej),'rp %.1e rd %.1e err %.1e'%h[-1][1:], 'accepted',naa,'rejected',nrej, flush=True)
