"""Local refinement after SCORE (reference README.md:63-67).

SCORE's convex relaxation gives an initial estimate; the reference's README hands it to a local
nonlinear least-squares solver (GTSAM in the paper) for the maximum-likelihood estimate on the
manifold.  This module is that next step, in 2-D and in 3-D: Gauss-Newton with Levenberg-Marquardt damping
on SE(d)^N x R^(d L) (3-D: steps in the tangent space, retraction R <- R Exp(omega), t <- t + v; class
``_Problem3D``), the first pose of the first chain held fixed (the gauge SCORE fixes too), over
exactly the factors SCORE reads (relative-pose measurements with the reference's chordal rotation cost,
gurobi_utils.py:504-526; ranges :449-501; landmark priors :433-446):

    F(theta, t, l) = sum_rel  kappa |t_j - t_i - R(theta_i) t_ij|^2 + tau |R(theta_j) - R(theta_i) R_ij|_F^2
                   + sum_rng  w (|p_a - p_b| - d_ab)^2  +  sum_prior w |l - l0|^2

The residuals and the sparse Jacobian are assembled vectorised on the host (NumPy / SciPy); the damped
normal equations (J'J + lambda I) step = -J'r -- block-tridiagonal 3 x 3 pose chains along every
robot's odometry, plus loop-closure and range couplings -- are solved on the GPU by the solver's
chain-preconditioned PCG through the C ABI (``score_linear_create`` / ``score_linear_solve``:
``k_factor`` factors the chains of J'J, ``k_prec_pre`` + ``k_spmv`` run the PCG, termination on the
device).  ``linear_solver="scipy"`` keeps the sparse-LU solve on the host: the reference the tests
compare the device path with, not a fallback (the default path raises without the HIP library).

``engine="native"`` (default) runs the WHOLE loop behind the C ABI (``score_refine_create`` /
``score_refine_run``, csrc/score_gn.hpp): per-measurement Jacobian blocks, J'J / J'r on a fixed pattern
and trial points are device kernels too, the host only steers the Levenberg-Marquardt iteration.
``engine="python"`` is the loop below (host Jacobians), kept as the readable twin the tests compare with.
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

    def retract(self, u, step):
        return u + step

    def chains(self):
        """One 3 x 3 chain per robot, node = pose (theta, x, y); the pinned pose is not an unknown."""
        lens = np.asarray(self.a["chain_len"], dtype=np.int64).copy()
        lens[0] -= 1
        lens = lens[lens > 0]
        return np.concatenate([[0], np.cumsum(lens)]), 3 * np.arange(self.Np - 1, dtype=np.int64)


def _hat(v: np.ndarray) -> np.ndarray:
    """[v]x for a stack of 3-vectors."""
    K = np.zeros(v.shape[:-1] + (3, 3))
    K[..., 0, 1], K[..., 0, 2] = -v[..., 2], v[..., 1]
    K[..., 1, 0], K[..., 1, 2] = v[..., 2], -v[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -v[..., 1], v[..., 0]
    return K


def so3_exp(w: np.ndarray) -> np.ndarray:
    """Rodrigues' formula for a stack of rotation vectors."""
    th = np.linalg.norm(w, axis=-1)
    K = _hat(w)
    small = th < 1e-8
    ths = np.where(small, 1.0, th)
    A = np.where(small, 1.0 - th ** 2 / 6.0, np.sin(ths) / ths)
    B = np.where(small, 0.5 - th ** 2 / 24.0, (1.0 - np.cos(ths)) / ths ** 2)
    return np.eye(3) + A[..., None, None] * K + B[..., None, None] * (K @ K)


class _Problem3D:
    """The 3-D RA-SLAM least-squares problem on SE(3)^N x R^(3 L) (the reference's model is dimension-generic,
    gurobi_utils.py:37-50).  The STATE is (R, t, lm): rotation matrices, translations, landmarks; a STEP lives in the
    tangent space, (omega, v) per free pose and a 3-vector per landmark, applied by the retraction
    R <- R Exp(omega), t <- t + v.  Columns of pose p >= 1: 6 (p - 1) .. +5 = [omega | v]; pose 0 is fixed."""

    def __init__(self, data):
        if data.dimension != 3:
            raise ValueError("_Problem3D: 3-D graphs only")
        a = graph_arrays(data)
        self.a = a
        self.Np, self.Nl = len(a["pose_names"]), len(a["landmark_names"])
        self.n = 6 * (self.Np - 1) + 3 * self.Nl
        self.bi, self.tj = a["rel_base"].astype(np.int64), a["rel_to"].astype(np.int64)
        self.tm, self.Rm = a["rel_t"], a["rel_R"]
        self.sk, self.st = np.sqrt(a["rel_kappa"]), np.sqrt(a["rel_tau"])
        self.ra, self.rb = a["rng_a"].astype(np.int64), a["rng_b"].astype(np.int64)
        self.dist, self.sw = a["rng_dist"], np.sqrt(a["rng_prec"])
        self.pl, self.pt, self.spw = a["lprior_lm"].astype(np.int64), a["lprior_t"].reshape(-1, 3), np.sqrt(a["lprior_prec"])

    def _point(self, v, t, lm):
        pose = v < self.Np
        out = np.empty((len(v), 3))
        out[pose] = t[v[pose]]
        out[~pose] = lm[v[~pose] - self.Np]
        return out

    def retract(self, state, step):
        R, t, lm = state
        w = step[: 6 * (self.Np - 1)].reshape(-1, 6)
        Rn, tn = R.copy(), t.copy()
        Rn[1:] = R[1:] @ so3_exp(w[:, :3])
        tn[1:] = t[1:] + w[:, 3:]
        return Rn, tn, lm + step[6 * (self.Np - 1):].reshape(-1, 3)

    def residuals(self, state, jac: bool = False):
        R, t, lm = state
        bi, tj = self.bi, self.tj
        ne = len(bi)
        rt = t[tj] - t[bi] - np.einsum("eij,ej->ei", R[bi], self.tm)
        rR = (R[tj] - R[bi] @ self.Rm).reshape(ne, 9)
        pa, pb = self._point(self.ra, t, lm), self._point(self.rb, t, lm)
        dv = pa - pb
        rho = np.linalg.norm(dv, axis=1)
        rp = lm[self.pl] - self.pt if len(self.pl) else np.zeros((0, 3))
        res = np.concatenate([(self.sk[:, None] * rt).ravel(), (self.st[:, None] * rR).ravel(), self.sw * (rho - self.dist),
                              (self.spw[:, None] * rp).ravel()])
        if not jac:
            return res
        rows, cols, vals = [], [], []

        def add(r, c, v):
            r, c, v = np.broadcast_arrays(np.asarray(r), np.asarray(c), np.asarray(v))
            r, c, v = r.ravel(), c.ravel(), v.ravel()
            keep = c >= 0
            rows.append(r[keep]); cols.append(c[keep]); vals.append(v[keep])

        NEG = -(10 ** 9)
        e = np.arange(ne)
        ci, cj = np.where(bi > 0, 6 * (bi - 1), NEG), np.where(tj > 0, 6 * (tj - 1), NEG)
        k3 = np.arange(3)
        r_t = 3 * e[:, None] + k3[None, :]
        add(r_t, cj[:, None] + 3 + k3[None, :], self.sk[:, None])
        add(r_t, ci[:, None] + 3 + k3[None, :], -self.sk[:, None])
        M = R[bi] @ _hat(self.tm)  # d r_t / d omega_i = R_i [tm]x
        for a_ in range(3):
            add(r_t, ci[:, None] + a_, self.sk[:, None] * M[:, :, a_])
        base = 3 * ne
        r_R = base + 9 * e[:, None] + np.arange(9)[None, :]
        for a_ in range(3):
            Ea = _hat(np.eye(3)[a_])
            add(r_R, cj[:, None] + a_, self.st[:, None] * (R[tj] @ Ea).reshape(ne, 9))
            add(r_R, ci[:, None] + a_, -self.st[:, None] * (R[bi] @ Ea @ self.Rm).reshape(ne, 9))
        base += 9 * ne
        nr = len(self.ra)
        r = base + np.arange(nr)
        g = dv / np.where(rho > 1e-12, rho, 1.0)[:, None]
        g[rho <= 1e-12] = 0.0

        def pcol(v):
            pose = v < self.Np
            return np.where(pose, np.where(v > 0, 6 * (v - 1) + 3, NEG), 6 * (self.Np - 1) + 3 * (v - self.Np))

        add(r[:, None], pcol(self.ra)[:, None] + k3[None, :], self.sw[:, None] * g)
        add(r[:, None], pcol(self.rb)[:, None] + k3[None, :], -self.sw[:, None] * g)
        base += nr
        npz = len(self.pl)
        add(base + 3 * np.arange(npz)[:, None] + k3[None, :], 6 * (self.Np - 1) + 3 * self.pl[:, None] + k3[None, :], self.spw[:, None])
        J = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(len(res), self.n))
        return res, J

    def cost(self, state) -> float:
        r = self.residuals(state)
        return float(r @ r)

    def initial_state(self, results):
        T = _stacked(results.poses, self.a["pose_names"], (4, 4))
        lm = _stacked(results.landmarks, self.a["landmark_names"], (3,)) if self.Nl else np.zeros((0, 3))
        return T[:, :3, :3].copy(), T[:, :3, 3].copy(), lm.copy()

    def chains(self):
        """3 x 3 chains of the preconditioner: per robot the omega blocks and the v blocks of its free poses."""
        lens = np.asarray(self.a["chain_len"], dtype=np.int64)
        ptr, first, p0 = [0], [], 0
        for L in lens:
            ps = np.arange(p0, p0 + L)
            ps = ps[ps > 0]
            for kind in range(2):
                if ps.size:
                    first.append(6 * (ps - 1) + 3 * kind)
                    ptr.append(ptr[-1] + ps.size)
            p0 += L
        return np.asarray(ptr, dtype=np.int64), (np.concatenate(first) if first else np.zeros(0, np.int64))


def _stacked(mapping, names, shape):
    """Values of ``mapping`` in the order of ``names`` as one array: without a Python loop when the mapping is
    an ArrayDict over the same names (what solve_score returns)."""
    if isinstance(mapping, compat.ArrayDict) and len(mapping) == len(names) and list(mapping.keys()) == list(names):
        return np.asarray(mapping.array, dtype=np.float64).reshape((len(names),) + shape)
    return np.array([np.asarray(mapping[nm], dtype=np.float64) for nm in names], dtype=np.float64).reshape((len(names),) + shape)


def _initial_point(prob: _Problem, results) -> np.ndarray:
    T = _stacked(results.poses, prob.a["pose_names"], (3, 3))
    th = np.arctan2(T[:, 1, 0], T[:, 0, 0])
    t = T[:, :2, 2].copy()
    lm = _stacked(results.landmarks, prob.a["landmark_names"], (2,))
    prob.pin = (float(th[0]), t[0].copy())
    return prob.pack(th, t, lm)


class _DeviceNormalEquations:
    """(J'J + lambda I) step = rhs on the GPU.  The pattern of J'J is fixed by the graph: it is taken once
    from the all-ones Jacobian (no cancellation can remove an entry), a linear-mode handle is created on it
    with one chain per robot (node = pose: theta, x, y), and every solve maps the current values onto it."""

    def __init__(self, prob: _Problem, J: sp.csr_matrix, lib_path: Optional[str], settings: Optional[dict]):
        from .solver import LinearSolver

        ones = J.copy()
        ones.data[:] = 1.0
        pat = (ones.T @ ones + sp.identity(prob.n, format="csr")).tocsr()
        pat.sort_indices()
        self.n = prob.n
        self.indptr, self.indices = pat.indptr.astype(np.int64), pat.indices.astype(np.int64)
        rows = np.repeat(np.arange(self.n, dtype=np.int64), np.diff(self.indptr))
        self.keys = rows * self.n + self.indices  # ascending: CSR with sorted indices
        self.diag = np.searchsorted(self.keys, np.arange(self.n, dtype=np.int64) * (self.n + 1))
        chain_ptr, node_first_col = prob.chains()
        self.solver = LinearSolver(pat, chain_ptr, node_first_col, 3, settings=settings, lib_path=lib_path)
        self._cache = None  # (indptr, indices) of the last J'J and its positions in the pattern
        self.pcg_iters = 0
        self.solves = 0

    def values(self, H: sp.csr_matrix) -> np.ndarray:
        c = self._cache
        if c is None or c[0].shape != H.indptr.shape or c[1].shape != H.indices.shape or \
                not (np.array_equal(c[0], H.indptr) and np.array_equal(c[1], H.indices)):
            rows = np.repeat(np.arange(self.n, dtype=np.int64), np.diff(H.indptr))
            pos = np.searchsorted(self.keys, rows * self.n + H.indices.astype(np.int64))
            if pos.size and (pos.max() >= self.keys.size or not np.array_equal(self.keys[pos], rows * self.n + H.indices)):
                raise RuntimeError("refine: J'J left its sparsity pattern")
            self._cache = c = (H.indptr.copy(), H.indices.copy(), pos)
        v = np.zeros(self.keys.size)
        if H.has_canonical_format:
            v[c[2]] = H.data
        else:
            np.add.at(v, c[2], H.data)  # duplicates summed
        return v

    def solve(self, H: sp.csr_matrix, lam: float, rhs: np.ndarray, rel_tol: float) -> np.ndarray:
        v = self.values(H)
        v[self.diag] += lam
        x, info = self.solver.solve(v, rhs, rel_tol=rel_tol, max_iters=4000)
        self.pcg_iters += info["iters"]
        self.solves += 1
        if not np.all(np.isfinite(x)):
            raise RuntimeError("refine: the device PCG returned a non-finite step")
        return x

    def close(self) -> None:
        self.solver.close()


def _refine_native(prob: _Problem, u0: np.ndarray, max_iters: int, tol: float, lib_path: Optional[str],
                   solver_settings: Optional[dict]):
    """The whole LM loop behind the C ABI (score_refine_*).  Returns (u, info dict)."""
    import ctypes as C

    from .native import score_graph_struct
    from .solver import ScoreRefineInfo, ScoreSettings, _f64p, load_library

    lib = load_library(lib_path)
    st = ScoreSettings()
    lib.score_default_settings(C.byref(st))
    for k, v in (solver_settings or {}).items():
        if not hasattr(st, k):
            raise ValueError(f"unknown solver setting {k}")
        setattr(st, k, v)
    g = score_graph_struct(prob.a)
    h = C.c_void_p()
    if lib.score_refine_create(C.byref(g), C.byref(st), C.byref(h)) != 0:
        raise RuntimeError(f"score_refine_create failed: {lib.score_last_error().decode()}")
    try:
        if isinstance(prob, _Problem3D):  # 3-D: [R (row-major) | t] per pose, landmarks x 3
            R, t, lm = u0
            poses_in = np.ascontiguousarray(np.concatenate([R.reshape(prob.Np, 9), t], axis=1), dtype=np.float64)
            lms_in = np.ascontiguousarray(lm, dtype=np.float64).reshape(-1, 3)
        else:
            th, t, lm = prob.split(u0)
            poses_in = np.ascontiguousarray(np.column_stack([th, t]), dtype=np.float64)
            lms_in = np.ascontiguousarray(lm, dtype=np.float64).reshape(-1, 2)
        poses_out, lms_out = np.empty_like(poses_in), np.empty((max(1, len(lms_in)), lms_in.shape[1]))
        info = ScoreRefineInfo()
        rc = lib.score_refine_run(h, poses_in.ctypes.data_as(_f64p), lms_in.ctypes.data_as(_f64p) if len(lms_in) else None,
                                  int(max_iters), float(tol), poses_out.ctypes.data_as(_f64p), lms_out.ctypes.data_as(_f64p),
                                  C.byref(info))
        if rc != 0:
            raise RuntimeError(f"score_refine_run failed: {lib.score_last_error().decode()}")
    finally:
        lib.score_refine_destroy(h)
    if isinstance(prob, _Problem3D):
        return (poses_out[:, :9].reshape(-1, 3, 3).copy(), poses_out[:, 9:12].copy(), lms_out[: len(lms_in)].copy()), info.as_dict()
    u = prob.pack(poses_out[:, 0], poses_out[:, 1:3], lms_out[: len(lms_in)])
    return u, info.as_dict()


def refine_estimate(data, results, max_iters: int = 50, tol: float = 1e-10, verbose: bool = False,
                    linear_solver: str = "device", lib_path: Optional[str] = None, solver_settings: Optional[dict] = None,
                    pcg_rel_tol: float = 1e-9, engine: str = "native"):
    """Refine a SCORE estimate (``SolverResults``) to a local minimiser of the RA-SLAM maximum-likelihood
    cost.  Returns ``(refined SolverResults, info)``; ``info`` holds the cost before / after, iterations,
    the final gradient norm and (device path) the PCG iterations spent in the linear solves."""
    if linear_solver not in ("device", "scipy"):
        raise ValueError("linear_solver must be 'device' or 'scipy'")
    if engine not in ("native", "python"):
        raise ValueError("engine must be 'native' or 'python'")
    if data.dimension == 3:
        prob = _Problem3D(data)
        u = prob.initial_state(results)
    else:
        prob = _Problem(data)
        u = _initial_point(prob, results)
    if engine == "native" and linear_solver == "device" and prob.n > 0:
        u, ni = _refine_native(prob, u, max_iters, tol, lib_path, solver_settings)
        info = {"cost_initial": ni["cost_initial"], "cost_final": ni["cost_final"], "iterations": ni["iterations"],
                "grad_inf": ni["grad_inf"], "linear_solver": "device", "engine": "native", "pcg_iters": ni["pcg_iters"],
                "linear_solves": ni["linear_solves"], "setup_ms": ni["setup_ms"], "solve_ms": ni["solve_ms"]}
        return _as_results(prob, u, results, ni["cost_final"]), info
    res, J = prob.residuals(u, jac=True)
    f = float(res @ res)
    f0 = f
    lam = 1e-6
    it = 0
    gnorm = np.inf
    dev = _DeviceNormalEquations(prob, J, lib_path, solver_settings) if (linear_solver == "device" and prob.n > 0) else None
    try:
        u, f, it, gnorm = _lm_loop(prob, u, res, J, f, lam, max_iters, tol, verbose, dev, pcg_rel_tol)
    finally:
        pcg = (dev.pcg_iters, dev.solves) if dev else (0, 0)
        if dev:
            dev.close()
    out = _as_results(prob, u, results, f)
    info = {"cost_initial": f0, "cost_final": f, "iterations": it, "grad_inf": gnorm, "linear_solver": linear_solver,
            "engine": "python", "pcg_iters": pcg[0], "linear_solves": pcg[1]}
    return out, info


def _as_results(prob, u, results, cost: float):
    if isinstance(prob, _Problem3D):
        R, t, lm = u
        T = np.tile(np.eye(4), (prob.Np, 1, 1))
        T[:, :3, :3] = R
        T[:, :3, 3] = t
        values = compat.VariableValues(3, compat.ArrayDict(prob.a["pose_names"], T), compat.ArrayDict(prob.a["landmark_names"], np.array(lm, dtype=np.float64)), None)
        return compat.SolverResults(variables=values, total_time=results.total_time, solved=True,
                                    pose_chain_names=results.pose_chain_names, solver_cost=cost, info=dict(results.info or {}))
    th, t, lm = prob.split(u)
    c, s = np.cos(th), np.sin(th)
    T = np.tile(np.eye(3), (prob.Np, 1, 1))
    T[:, 0, 0] = c; T[:, 0, 1] = -s; T[:, 1, 0] = s; T[:, 1, 1] = c
    T[:, :2, 2] = t
    values = compat.VariableValues(2, compat.ArrayDict(prob.a["pose_names"], T), compat.ArrayDict(prob.a["landmark_names"], lm.copy()), None)
    return compat.SolverResults(variables=values, total_time=results.total_time, solved=True,
                                pose_chain_names=results.pose_chain_names, solver_cost=cost, info=dict(results.info or {}))


def _lm_loop(prob, u, res, J, f, lam, max_iters, tol, verbose, dev, pcg_rel_tol):
    it = 0
    gnorm = np.inf
    for it in range(1, max_iters + 1):
        g = J.T @ res
        gnorm = float(np.abs(g).max()) if g.size else 0.0
        if gnorm <= tol * max(1.0, f):
            break
        H = (J.T @ J).tocsr() if dev else (J.T @ J).tocsc()
        accepted = False
        for _ in range(12):
            try:
                if dev:
                    step = dev.solve(H, lam, -g, pcg_rel_tol)
                else:
                    step = spla.splu((H + lam * sp.identity(prob.n, format="csc")).tocsc()).solve(-g)
            except RuntimeError:
                lam *= 10.0
                continue
            un = prob.retract(u, step)
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
    return u, f, it, gnorm
