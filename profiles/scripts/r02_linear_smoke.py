import sys, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla, time
sys.path.insert(0,'.')
from score_amd.solver import LinearSolver
TW=None if len(sys.argv) < 2 else sys.argv[1]
rng=np.random.default_rng(0)
# 3 chains of 50 nodes, block 3, + 4 extra scalar unknowns coupled randomly
N,bs,nch=50,3,3
n=N*bs*nch+4
rows,cols,vals=[],[],[]
def addblk(i,j,B):
    for a in range(B.shape[0]):
        for b in range(B.shape[1]):
            rows.append(i+a); cols.append(j+b); vals.append(B[a,b])
for c in range(nch):
    base=c*N*bs
    for k in range(N):
        M=rng.normal(size=(bs,bs)); addblk(base+k*bs, base+k*bs, M@M.T+3*np.eye(bs))
        if k>0:
            B=0.8*rng.normal(size=(bs,bs)); addblk(base+k*bs, base+(k-1)*bs, B); addblk(base+(k-1)*bs, base+k*bs, B.T)
for e in range(4):
    i=N*bs*nch+e
    rows.append(i); cols.append(i); vals.append(5.0)
    for _ in range(6):
        j=int(rng.integers(0,N*bs*nch)); v=0.3*rng.normal()
        rows += [i,j]; cols += [j,i]; vals += [v,v]
K=sp.csr_matrix((vals,(rows,cols)),shape=(n,n)); K.sum_duplicates()
K = K + sp.identity(n)*10  # make SPD comfortably
K=K.tocsr(); K.sort_indices()
w=np.linalg.eigvalsh(K.toarray()); print('min eig', w.min())
cp=np.arange(nch+1)*N; nfc=np.concatenate([c*N*bs+np.arange(N)*bs for c in range(nch)])
ls=LinearSolver(K, cp, nfc, bs, lib_path=TW)
b=rng.normal(size=n)
x,info=ls.solve(K.data,b,rel_tol=1e-10,residual=True)
xr=spla.spsolve(K.tocsc(),b)
print(info, np.abs(x-xr).max()/np.abs(xr).max())
K2=K.copy(); K2.data=K.data*2.0
x,info=ls.solve(K2.data,b,rel_tol=1e-10,residual=True); print(info, np.abs(x-xr/2).max())
x,info=ls.solve(K2.data,np.zeros(n)); print(info, np.abs(x).max())
ls.close()
