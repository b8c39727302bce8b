"""ORACLE / TEST INFRASTRUCTURE -- CPU baseline "C1" of BASELINE.md section 2.

A single-threaded SciPy restatement of the operator-splitting conic solver in its
DIRECT form (the published OSQP / SCS-direct scheme: equilibrate, factor the
quasi-definite KKT system once per penalty value with a sparse direct solver,
one triangular solve pair per iteration), applied to the standard-form program
`score_amd.assemble` builds from a FactorGraphData:

    minimise 1/2 x'Px + q'x + c0   s.t.  A x + s = b,  s in SOC(d_1) x ... x SOC(d_k)

`north_star` names "CVXPY -> SCS" as the reference CPU path; neither package (nor
gurobipy, which the reference actually calls at score/solve_score.py:76) exists in
this image, so this is the closest stand-in that can be timed on the GPU box's host
cores: same splitting, same over-relaxation, same stopping rule and epsilon as the
HIP solver, exact KKT solves (SuperLU) instead of preconditioned CG.

Only bench.py's cpu_baseline leg and tests/ import this module; the product never does.
"""
from __future__ import annotations

import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def _ruiz(P, A, soc_dims, iters=10):
    """Ruiz equilibration of [[P, A'], [A, 0]] with one scale per cone (as the HIP solver's setup)."""
    n, m = P.shape[0], A.shape[0]
    D, E = np.ones(n), np.ones(m)
    P, A = P.tocsr().copy(), A.tocsr().copy()
    starts = np.concatenate([[0], np.cumsum(soc_dims)])[:-1] if len(soc_dims) else np.zeros(0, int)
    for _ in range(iters):
        cn = np.maximum(abs(P).max(axis=0).toarray().ravel(), abs(A).max(axis=0).toarray().ravel() if m else 0.0)
        d = np.where(cn > 1e-12, 1.0 / np.sqrt(np.maximum(cn, 1e-300)), 1.0)
        if m:
            rn = abs(A).max(axis=1).toarray().ravel()
            if len(soc_dims):
                rn = np.repeat(np.maximum.reduceat(rn, starts), soc_dims)
            e = np.where(rn > 1e-12, 1.0 / np.sqrt(np.maximum(rn, 1e-300)), 1.0)
        else:
            e = np.ones(0)
        Dm, Em = sp.diags(d), sp.diags(e)
        P = (Dm @ P @ Dm).tocsr()
        A = (Em @ A @ Dm).tocsr()
        D *= d
        E *= e
    return P, A, D, E


def _proj_soc_blocks(v, dim):
    """Projection of every row block of v (k x dim) onto SOC(dim), vectorised."""
    t, z = v[:, 0], v[:, 1:]
    nz = np.sqrt(np.einsum("ij,ij->i", z, z))
    out = np.zeros_like(v)
    inside = nz <= t
    polar = nz <= -t
    mid = ~(inside | polar)
    out[inside] = v[inside]
    a = 0.5 * (t[mid] + nz[mid])
    out[mid, 0] = a
    out[mid, 1:] = (a / nz[mid])[:, None] * z[mid]
    return out


def solve(qp, eps=1e-7, max_iters=20000, rho=0.1, sigma=1e-6, alpha=1.8, check=25, rho_interval=100, rho_tol=5.0,
          time_limit=None):
    """ADMM to the HIP solver's stopping rule.  Returns dict(x, y, s, iters, seconds, factorizations,
    solved, pobj, res_pri, res_dual)."""
    t0 = time.perf_counter()
    soc = np.asarray(qp.soc_dims, dtype=np.int64)
    assert qp.z == 0 and (len(soc) == 0 or np.all(soc == soc[0])), "SCORE programs: SOC blocks of one size"
    dim = int(soc[0]) if len(soc) else 1
    P, A, D, E = _ruiz(qp.P, qp.A, soc)
    q, b = qp.q * D, qp.b * E
    n, m = P.shape[0], A.shape[0]
    At = A.T.tocsr()
    AtA = (At @ A).tocsc() if m else sp.csc_matrix((n, n))
    Pc = (P + sigma * sp.identity(n)).tocsc()
    nfac = 0

    def factor(r):
        nonlocal nfac
        nfac += 1
        return spla.splu((Pc + r * AtA).tocsc(), permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0)

    lu = factor(rho)
    x = np.zeros(n); s = np.zeros(m); y = np.zeros(m)
    bn = np.abs(qp.b).max() if m else 0.0
    it, solved = 0, False
    info = {}
    next_rho = rho_interval
    while it < max_iters:
        for _ in range(check):
            xt = lu.solve(sigma * x - q + At @ (rho * (b - s) - y))
            x = alpha * xt + (1 - alpha) * x
            v = alpha * (b - A @ xt) + (1 - alpha) * s
            s = _proj_soc_blocks((v - y / rho).reshape(-1, dim), dim).ravel() if m else s
            y = y + rho * (s - v)
            it += 1
        # unscaled residuals, the HIP solver's tests (score_driver.hpp check())
        Ax = A @ x
        rp = (Ax + s - b) / E if m else np.zeros(0)
        Px = P @ x
        Aty = At @ y
        rd = (Px + q + Aty) / D
        rp_u = np.abs(rp).max() if m else 0.0
        rd_u = np.abs(rd).max()
        pn = max(np.abs(Ax / E).max() if m else 0.0, np.abs(s / E).max() if m else 0.0, bn)
        dn = np.abs(Aty / D).max()
        xPx, qx = float(x @ Px), float(q @ x)
        gap = abs(float(x @ (Px + q + Aty)) + float(s @ y) - float(y @ (Ax + s - b)))
        ok = (rp_u <= eps + eps * pn and rd_u <= eps + eps * dn
              and gap <= eps + eps * max(abs(xPx), abs(qx), abs(float(b @ y)) if m else 0.0))
        info = dict(res_pri=float(rp_u), res_dual=float(rd_u), gap=float(gap), pobj=0.5 * xPx + qx + float(qp.c0))
        if ok:
            solved = True
            break
        if it >= next_rho and m:
            next_rho = it + rho_interval
            pns = max(np.abs(Ax).max(), np.abs(s).max(), np.abs(b).max(), 1e-30)
            dns = max(np.abs(Px).max(), np.abs(Aty).max(), np.abs(q).max(), 1e-30)
            ratio = np.sqrt((np.abs(Ax + s - b).max() / pns) / (max(np.abs(Px + q + Aty).max(), 1e-30) / dns))
            nr = min(1e6, max(1e-6, rho * ratio))
            if nr > rho_tol * rho or nr < rho / rho_tol:
                rho = nr
                lu = factor(rho)
        if time_limit is not None and time.perf_counter() - t0 > time_limit:
            break
    out = dict(x=x * D, y=y * E, s=s / E if m else s, iters=it, seconds=time.perf_counter() - t0, factorizations=nfac,
               solved=solved, rho=rho)
    out.update(info)
    return out
