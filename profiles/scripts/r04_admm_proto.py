"""Scratch ADMM prototype (numpy/scipy) to choose algorithm parameters."""
import time, sys, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
from score_amd.io import load_pyfg_pickle
from score_amd.assemble import assemble
from oracle import score_oracle as so

def proj_soc_batch(V):
    # V: (nc, dim)
    t = V[:,0]; z = V[:,1:]; nz = np.linalg.norm(z, axis=1)
    out = V.copy()
    zero = nz <= -t
    out[zero] = 0
    mid = (~zero) & (nz > t)
    a = 0.5*(t[mid]+nz[mid])
    out[mid,0] = a
    out[mid,1:] = (a/nz[mid])[:,None]*z[mid]
    return out

def ruiz(P, A, soc_dim, iters=10):
    n = P.shape[0]; m = A.shape[0]
    D = np.ones(n); E = np.ones(m)
    Pc = P.copy(); Ac = A.copy()
    for _ in range(iters):
        cn = np.maximum(np.abs(Pc).max(axis=0).toarray().ravel(), np.abs(Ac).max(axis=0).toarray().ravel() if m else 0)
        rn = np.abs(Ac).max(axis=1).toarray().ravel() if m else np.zeros(0)
        # uniform within cone
        if m:
            rn = rn.reshape(-1, soc_dim).max(axis=1).repeat(soc_dim)
        d = 1/np.sqrt(np.where(cn>0, cn, 1)); e = 1/np.sqrt(np.where(rn>0, rn, 1))
        Dm = sp.diags(d); Em = sp.diags(e)
        Pc = Dm@Pc@Dm; Ac = Em@Ac@Dm
        D *= d; E *= e
    return D, E, Pc.tocsr(), Ac.tocsr()

def admm(qp, rho=1.0, sigma=1e-6, alpha=1.6, max_iter=20000, eps=1e-6, scale=True, adapt=True, verbose=True, xref=None, check=25):
    P, A, q, b = qp.P, qp.A, qp.q, qp.b
    n, m = qp.n, qp.m; dim = int(qp.soc_dims[0])
    if scale:
        D, E, Ps, As = ruiz(P, A, dim)
        qs = D*q; 
        c = 1/max(np.mean(np.abs(Ps).max(axis=0).toarray()), np.max(np.abs(qs)), 1e-12) if False else 1.0
        bs = E*b
    else:
        D = np.ones(n); E = np.ones(m); Ps, As, qs, bs = P, A, q, b
    x = np.zeros(n); s = np.zeros(m); y = np.zeros(m)
    def factor(rho):
        K = (Ps + sigma*sp.identity(n) + rho*(As.T@As)).tocsc()
        return spla.splu(K)
    lu = factor(rho)
    t0=time.time(); nfac=1
    hist=[]
    for it in range(1, max_iter+1):
        rhs = sigma*x - qs + As.T@(rho*(bs - s) - y)
        xt = lu.solve(rhs)
        Axt = As@xt
        xn = alpha*xt + (1-alpha)*x
        v = alpha*(bs - Axt) + (1-alpha)*s
        w = v - y/rho
        sn = proj_soc_batch(w.reshape(-1,dim)).ravel()
        y = y + rho*(sn - v)
        x = xn; s = sn
        if it % check == 0:
            # unscaled residuals
            xu = D*x; su = s/E; yu = E*y
            rp = np.max(np.abs(A@xu + su - b)); 
            Px = P@xu; Aty = A.T@yu
            rd = np.max(np.abs(Px + q + Aty))
            pn = max(np.max(np.abs(A@xu)), np.max(np.abs(su)), np.max(np.abs(b)))
            dn = max(np.max(np.abs(Px)), np.max(np.abs(Aty)), np.max(np.abs(q)))
            obj = 0.5*xu@Px + q@xu + qp.c0
            err = np.max(np.abs(xu-xref))/np.max(np.abs(xref)) if xref is not None else np.nan
            hist.append((it, rp, rd, obj, err))
            if verbose and it % (check*8) == 0:
                print(f"it {it:6d} rp {rp:.2e} rd {rd:.2e} obj {obj:.8f} rho {rho:.3g} relerr {err:.2e}")
            if rp <= eps*(1+pn) and rd <= eps*(1+dn):
                break
            if adapt and it % (check*4) == 0:
                # scaled residual ratio (OSQP)
                rps = np.max(np.abs(As@x + s - bs)); rds = np.max(np.abs(Ps@x + qs + As.T@y))
                pns = max(np.max(np.abs(As@x)), np.max(np.abs(s)), np.max(np.abs(bs)),1e-12)
                dns = max(np.max(np.abs(Ps@x)), np.max(np.abs(As.T@y)), np.max(np.abs(qs)),1e-12)
                new = rho*np.sqrt((rps/pns)/(max(rds,1e-15)/dns))
                new = min(max(new, 1e-6), 1e6)
                if new > 5*rho or new < rho/5:
                    rho = new; lu = factor(rho); nfac+=1
    return D*x, s/E, E*y, dict(iters=it, time=time.time()-t0, nfac=nfac, hist=hist, rho=rho)

if __name__ == '__main__':
    fg = load_pyfg_pickle('/root/reference/examples/manhattan/factor_graph.pickle')
    relax = sys.argv[1] if len(sys.argv)>1 else 'SOCP'
    mdl = assemble(fg, relax)
    rp_,u,info = so.newton_solve(fg, tol=1e-14)
    vals = so.reduced_to_values(rp_,u,relax)
    xm = np.zeros(mdl.n_model)
    for i,nm in enumerate(mdl.pose_names): xm[i*6:(i+1)*6] = vals['poses'][nm].ravel()
    for i,nm in enumerate(mdl.landmark_names): xm[mdl.lm_base+i*2:mdl.lm_base+i*2+2]=vals['landmarks'][nm]
    for i,k in enumerate(mdl.range_keys): xm[mdl.rng_base+i*mdl.rng_width:mdl.rng_base+(i+1)*mdl.rng_width]=vals['dists'][k]
    xref = mdl.reduce(xm)
    for kw in [dict(scale=False, adapt=False, rho=1.0), dict(scale=False, adapt=True), dict(scale=True, adapt=True, rho=0.1)]:
        print(kw)
        x,s,y,inf = admm(mdl.qp, xref=xref, **kw)
        print('  -> iters', inf['iters'], 'time', inf['time'], 'nfac', inf['nfac'], 'rho', inf['rho'], 'final', inf['hist'][-1])
