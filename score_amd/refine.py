"""Local refinement after SCORE (reference README.md:63-67).

SCORE's convex relaxation gives an initial estimate; the reference's README hands it to a local
nonlinear least-squares solver (GTSAM in the paper) for the maximum-likelihood estimate on the
manifold.  This module is that next step for 2-D graphs: Gauss-Newton with Levenberg-Marquardt damping
on SE(2)^N x R^(2 L), the first pose of the first chain held fixed (the gauge SCORE fixes too), over
exactly the factors SCORE reads (relative-pose measurements with the reference's chordal rotation cost,
gurobi_utils.py:504-526; ranges :449-501; landmark priors :433-446):

    F(theta, t, l) = sum_rel  kappa |t_j - t_i - R(theta_i) t_ij|^2 + tau |R(theta_j) - R(theta_i) R_ij|_F^2
                   + sum_rng  w (|p_a - p_b| - d_ab)^2  +  sum_prior w |l - l0|^2

Host-side (NumPy / SciPy sparse): the Jacobian is assembled vectorised, the normal equations are solved
by sparse Cholesky-like LU.  The GPU path of this round stops at the SCORE estimate; the normal
equations here have the structure of the polish's Newton systems (block-tridiagonal pose chains + range
couplings), so they are the next candidate for the chain-preconditioned PCG on the device.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from . import compat
from .native import graph_arrays


class _Problem:
    """Residuals and sparse Jacobian of the 2-D RA-SLAM least-squares problem in the minimal
    parametrisation u = [theta_1.., x_1, y_1.. | landmarks]; pose 0 (first pose of chain 0) is fixed."""

    def __init__(self, data):
        if data.dimension != 2:
            raise ValueError("refine_estimate: 2-D graphs only")
        a = graph_arrays(data)
        self.a = a
        self.Np, self.Nl = len(a["pose_names"]), len(a["landmark_names"])
        self.n = 3 * (self.Np - 1) + 2 * self.Nl
        self.bi, self.tj = a["rel_base"].astype(np.int64), a["rel_to"].astype(np.int64)
        self.tm, self.Rm = a["rel_t"], a["rel_R"]
        self.sk, self.st = np.sqrt(a["rel_kappa"]), np.sqrt(a["rel_tau"])
        self.ra, self.rb = a["rng_a"].astype(np.int64), a["rng_b"].astype(np.int64)
        self.dist, self.sw = a["rng_dist"], np.sqrt(a["rng_prec"])
        self.pl, self.pt, self.spw = a["lprior_lm"].astype(np.int64), a["lprior_t"], np.sqrt(a["lprior_prec"])
        self.pin = (0.0, np.zeros(2))

    # ---- packing ----
    def split(self, u) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        th = np.concatenate([[self.pin[0]], u[0 : 3 * (self.Np - 1) : 3]])
        t = np.vstack([self.pin[1][None, :], np.stack([u[1 : 3 * (self.Np - 1) : 3], u[2 : 3 * (self.Np - 1) : 3]], axis=1)])
        lm = u[3 * (self.Np - 1) :].reshape(-1, 2)
        return th, t, lm

    def pack(self, th, t, lm) -> np.ndarray:
        u = np.empty(self.n)
        u[0 : 3 * (self.Np - 1) : 3] = th[1:]
        u[1 : 3 * (self.Np - 1) : 3] = t[1:, 0]
        u[2 : 3 * (self.Np - 1) : 3] = t[1:, 1]
        u[3 * (self.Np - 1) :] = lm.ravel()
        return u

    def _col_pose(self, p):  # first column of pose p (theta, x, y); -1 for the fixed pose
        return np.where(p > 0, 3 * (p - 1), -1)

    def _point(self, v, t, lm):
        pose = v < self.Np
        out = np.empty((len(v), 2))
        out[pose] = t[v[pose]]
        out[~pose] = lm[v[~pose] - self.Np]
        return out

    def _col_point(self, v):  # column of x of a range endpoint
        pose = v < self.Np
        return np.where(pose, np.where(v > 0, 3 * (v - 1) + 1, -1), 3 * (self.Np - 1) + 2 * (v - self.Np))

    # ---- residuals / Jacobian ----
    def residuals(self, u, jac: bool = False):
        th, t, lm = self.split(u)
        c, s = np.cos(th), np.sin(th)
        bi, tj = self.bi, self.tj
        ne = len(bi)
        ci, si, cj, sj = c[bi], s[bi], c[tj], s[tj]
        tm, Rm = self.tm, self.Rm
        # translation: t_j - t_i - R_i tm
        rt = t[tj] - t[bi] - np.stack([ci * tm[:, 0] - si * tm[:, 1], si * tm[:, 0] + ci * tm[:, 1]], axis=1)
        # rotation (chordal): R_j - R_i Rm, the four entries
        Ri = np.stack([ci, -si, si, ci], axis=1).reshape(ne, 2, 2)
        Rj = np.stack([cj, -sj, sj, cj], axis=1).reshape(ne, 2, 2)
        rr = (Rj - Ri @ Rm).reshape(ne, 4)
        pa, pb = self._point(self.ra, t, lm), self._point(self.rb, t, lm)
        dv = pa - pb
        rho = np.sqrt(np.einsum("ij,ij->i", dv, dv))
        rg = rho - self.dist
        rp = lm[self.pl] - self.pt if len(self.pl) else np.zeros((0, 2))
        res = np.concatenate([(self.sk[:, None] * rt).ravel(), (self.st[:, None] * rr).ravel(), self.sw * rg,
                              (self.spw[:, None] * rp).ravel()])
        if not jac:
            return res
        rows, cols, vals = [], [], []

        def add(r, cidx, v):
            keep = cidx >= 0
            rows.append(r[keep]); cols.append(cidx[keep]); vals.append(v[keep])

        e = np.arange(ne)
        coli, colj = self._col_pose(bi), self._col_pose(tj)
        # translation rows 2e, 2e+1
        for k in range(2):
            r = 2 * e + k
            add(r, np.where(colj >= 0, colj + 1 + k, -1), self.sk)
            add(r, np.where(coli >= 0, coli + 1 + k, -1), -self.sk)
        dRt = np.stack([-si * tm[:, 0] - ci * tm[:, 1], ci * tm[:, 0] - si * tm[:, 1]], axis=1)  # d(R_i tm)/dtheta_i
        for k in range(2):
            add(2 * e + k, coli, -self.sk * dRt[:, k])
        # rotation rows
        base = 2 * ne
        dRi = np.stack([-si, -ci, ci, -si], axis=1).reshape(ne, 2, 2)
        dRj = np.stack([-sj, -cj, cj, -sj], axis=1).reshape(ne, 2, 2)
        dri = -(dRi @ Rm).reshape(ne, 4)
        drj = dRj.reshape(ne, 4)
        for k in range(4):
            add(base + 4 * e + k, coli, self.st * dri[:, k])
            add(base + 4 * e + k, colj, self.st * drj[:, k])
        # ranges
        base += 4 * ne
        nr = len(self.ra)
        r = base + np.arange(nr)
        safe = np.where(rho > 1e-12, rho, 1.0)
        g = dv / safe[:, None]
        g[rho <= 1e-12] = 0.0
        ca, cb = self._col_point(self.ra), self._col_point(self.rb)
        for k in range(2):
            add(r, np.where(ca >= 0, ca + k, -1), self.sw * g[:, k])
            add(r, np.where(cb >= 0, cb + k, -1), -self.sw * g[:, k])
        base += nr
        npz = len(self.pl)
        for k in range(2):
            add(base + 2 * np.arange(npz) + k, 3 * (self.Np - 1) + 2 * self.pl + k, self.spw)
        J = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(len(res), self.n))
        return res, J

    def cost(self, u) -> float:
        r = self.residuals(u)
        return float(r @ r)


def _initial_point(prob: _Problem, results) -> np.ndarray:
    names = prob.a["pose_names"]
    th = np.empty(prob.Np)
    t = np.empty((prob.Np, 2))
    for i, nm in enumerate(names):
        T = results.poses[nm]
        th[i] = np.arctan2(T[1, 0], T[0, 0])
        t[i] = T[:2, 2]
    lm = np.array([results.landmarks[nm] for nm in prob.a["landmark_names"]]).reshape(-1, 2)
    prob.pin = (float(th[0]), t[0].copy())
    return prob.pack(th, t, lm)


def refine_estimate(data, results, max_iters: int = 50, tol: float = 1e-10, verbose: bool = False):
    """Refine a SCORE estimate (``SolverResults``) to a local minimiser of the RA-SLAM maximum-likelihood
    cost.  Returns ``(refined SolverResults, info)``; ``info`` holds the cost before / after, iterations and
    the final gradient norm."""
    prob = _Problem(data)
    u = _initial_point(prob, results)
    res, J = prob.residuals(u, jac=True)
    f = float(res @ res)
    f0 = f
    lam = 1e-6
    it = 0
    gnorm = np.inf
    for it in range(1, max_iters + 1):
        g = J.T @ res
        gnorm = float(np.abs(g).max()) if g.size else 0.0
        if gnorm <= tol * max(1.0, f):
            break
        H = (J.T @ J).tocsc()
        accepted = False
        for _ in range(12):
            try:
                step = spla.splu((H + lam * sp.identity(prob.n, format="csc")).tocsc()).solve(-g)
            except RuntimeError:
                lam *= 10.0
                continue
            un = u + step
            fn = prob.cost(un)
            if fn < f:
                accepted = True
                break
            lam *= 10.0
        if not accepted:
            break
        dec = f - fn
        u, f = un, fn
        lam = max(lam * 0.1, 1e-12)
        res, J = prob.residuals(u, jac=True)
        if verbose:
            print(f"  refine it {it}: cost {f:.9g} |g| {gnorm:.3e} lambda {lam:.1e}")
        if dec <= 1e-14 * max(1.0, f):
            break
    th, t, lm = prob.split(u)
    c, s = np.cos(th), np.sin(th)
    T = np.tile(np.eye(3), (prob.Np, 1, 1))
    T[:, 0, 0] = c; T[:, 0, 1] = -s; T[:, 1, 0] = s; T[:, 1, 1] = c
    T[:, :2, 2] = t
    values = compat.VariableValues(2, compat.ArrayDict(prob.a["pose_names"], T), compat.ArrayDict(prob.a["landmark_names"], lm.copy()), None)
    out = compat.SolverResults(variables=values, total_time=results.total_time, solved=True,
                               pose_chain_names=results.pose_chain_names, solver_cost=f, info=dict(results.info or {}))
    info = {"cost_initial": f0, "cost_final": f, "iterations": it, "grad_inf": gnorm}
    return out, info
