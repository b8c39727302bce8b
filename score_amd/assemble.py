"""FactorGraphData -> standard-form conic QP for the SCORE relaxation.

This is the vectorised counterpart of the reference's model construction
(score/utils/gurobi_utils.py:173-187 ``initialize_model``):

    minimise   1/2 x'Px + q'x + c0
    subject to A x + s = b,   s in {0}^z x SOC(d_1) x ... x SOC(d_k)

Column layout ("model space", mirrors the Gurobi MVar shapes):
  * pose p (poses numbered chain by chain, gurobi_utils.py:233-246): the
    d x (d+1) matrix [R | t] row-major, column  p*d*(d+1) + k*(d+1) + j
    holds R[k, j] for j < d and t[k] for j == d;
  * landmark l (:249-258): d columns;
  * range variable r (:261-310): one column d_ij (SOCP, lb = 0) or d columns
    r_ij (QCQP).
Objective terms (:358-526) are assembled as weighted least-squares rows
``w * (J x - c)^2`` so that P = 2 J'WJ, q = -2 J'Wc, c0 = c'Wc (the SOCP range
cost keeps its constant ``w*dist^2`` exactly like :487).
The pinned first pose (:181-183, :316-333) is eliminated from the unknowns
("solver space" = model space minus the d(d+1) pinned columns); this is the
same feasible set as the d(d+1) equality rows Gurobi receives.
Cones (:336-352): SOCP  (d_ij, t_i - t_j) in SOC(d+1); QCQP (1, r_ij) in
SOC(d+1).  The SOCP bound d_ij >= 0 (:290-293) is implied by the cone.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import scipy.sparse as sp

SOCP_RELAXATION = "SOCP"
QCQP_RELAXATION = "QCQP"
ACCEPTABLE_RELAXATIONS = [SOCP_RELAXATION, QCQP_RELAXATION]


def _dot(a: np.ndarray, b: np.ndarray) -> float:
    """Inner product that stays out of BLAS.  A threaded BLAS wakes its whole worker pool for a
    long vector, and those workers then spin for ~100 ms on the cores the native setup threads
    of the next `score_create` want (measured on the MI355X host: +80..130 ms per problem)."""
    return float(np.einsum("i,i->", a, b))


def check_valid_relaxation(relaxation: str) -> None:
    """gurobi_utils.py:139-144."""
    if relaxation not in ACCEPTABLE_RELAXATIONS:
        raise ValueError(
            f"Relaxation {relaxation} is not supported. "
            f"Acceptable relaxations are {ACCEPTABLE_RELAXATIONS}"
        )


def check_dimension(value) -> None:
    """gurobi_utils.py:37-50."""
    if not isinstance(value, (int, np.integer)) or isinstance(value, bool):
        raise ValueError(f"{value} is not an int")
    if value not in (2, 3):
        raise ValueError(f"Value {value} is not 2 or 3")


@dataclass
class ConicQP:
    """min 1/2 x'Px + q'x + c0  s.t.  Ax + s = b, s in {0}^z x prod SOC."""

    P: sp.csr_matrix  # full symmetric, n x n
    q: np.ndarray
    c0: float
    A: sp.csr_matrix  # m x n
    b: np.ndarray
    z: int  # leading zero-cone rows
    soc_dims: np.ndarray  # int32, sum == m - z
    # block-tridiagonal preconditioner hint: chain c = nodes
    # chain_ptr[c]..chain_ptr[c+1]-1, node j = columns node_cols[j*bs:(j+1)*bs]
    chain_ptr: np.ndarray = field(default_factory=lambda: np.zeros(1, np.int32))
    node_cols: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    block_size: int = 0
    # row-replication hint (include/score_hip.h, score_problem::rep_d / rep_n): the first rep_d * rep_n
    # columns are rep_d replicas of rep_n unknowns that only the cones couple; 0 = no hint
    rep_d: int = 0
    rep_n: int = 0

    @property
    def n(self) -> int:
        return self.P.shape[0]

    @property
    def m(self) -> int:
        return self.A.shape[0]

    def objective(self, x: np.ndarray) -> float:
        return 0.5 * _dot(x, self.P @ x) + _dot(self.q, x) + float(self.c0)


@dataclass
class ScoreModel:
    """The assembled problem plus the maps needed to read a solution back."""

    dim: int
    relaxation: str
    qp: ConicQP
    n_model: int  # columns in model space (with the pinned pose)
    free_cols: np.ndarray  # model-space column of each solver-space column
    fixed_cols: np.ndarray
    fixed_vals: np.ndarray
    pose_names: List[str]
    landmark_names: List[str]
    range_keys: List[Tuple[str, str]]
    lm_base: int
    rng_base: int
    rng_width: int  # 1 (SOCP) or d (QCQP)
    # per range measurement: first translation column / stride of both end points (model space)
    # and the measured distance
    range_ends: Optional[np.ndarray] = None  # (Nr, 4) int64: ta, sa, tb, sb
    range_dist: Optional[np.ndarray] = None
    pose_chain_names: Optional[list] = None  # FactorGraphData.get_pose_chain_names(), when the builder had it at hand

    def expand(self, x_solver: np.ndarray) -> np.ndarray:
        """solver space -> model space (re-inserts the pinned pose)."""
        x = np.empty(self.n_model, dtype=np.float64)
        x[self.free_cols] = x_solver
        x[self.fixed_cols] = self.fixed_vals
        return x

    def reduce(self, x_model: np.ndarray) -> np.ndarray:
        return np.ascontiguousarray(x_model[self.free_cols])

    def pose_blocks(self, x_model: np.ndarray) -> np.ndarray:
        """(N_p, d, d+1) array of [R | t]."""
        d = self.dim
        return x_model[: len(self.pose_names) * d * (d + 1)].reshape(-1, d, d + 1)

    def landmark_block(self, x_model: np.ndarray) -> np.ndarray:
        d = self.dim
        return x_model[self.lm_base : self.lm_base + len(self.landmark_names) * d].reshape(-1, d)

    def range_block(self, x_model: np.ndarray) -> np.ndarray:
        return x_model[self.rng_base :].reshape(-1, self.rng_width)


def _check_unique(names: List[str], what: str) -> None:
    """gurobi_utils.py:62-80 duplicate-variable guards."""
    seen = set()
    for nm in names:
        if nm in seen:
            raise ValueError(f"Variable name {nm} already exists in {what}")
        seen.add(nm)


def _pose_meas_arrays(meas: list, pose_idx: Dict[str, int], d: int):
    from operator import attrgetter

    ne = len(meas)
    bi = np.fromiter(map(pose_idx.__getitem__, map(attrgetter("base_pose"), meas)), dtype=np.int64, count=ne)
    tj = np.fromiter(map(pose_idx.__getitem__, map(attrgetter("to_pose"), meas)), dtype=np.int64, count=ne)
    kap = np.fromiter(map(attrgetter("translation_precision"), meas), dtype=np.float64, count=ne)
    tau = np.fromiter(map(attrgetter("rotation_precision"), meas), dtype=np.float64, count=ne)
    tm = np.empty((ne, d))
    Rm = np.empty((ne, d, d))
    m0 = meas[0]
    if d == 2 and all(hasattr(m0, a) for a in ("x", "y", "theta")):
        # PyFactorGraph's PoseMeasurement2D stores (x, y, theta); translation_vector and
        # rotation_matrix are properties derived from them -- derive them for all edges at once
        tm[:, 0] = np.fromiter(map(attrgetter("x"), meas), dtype=np.float64, count=ne)
        tm[:, 1] = np.fromiter(map(attrgetter("y"), meas), dtype=np.float64, count=ne)
        th = np.fromiter(map(attrgetter("theta"), meas), dtype=np.float64, count=ne)
        cs, sn = np.cos(th), np.sin(th)
        Rm[:, 0, 0] = cs; Rm[:, 0, 1] = -sn
        Rm[:, 1, 0] = sn; Rm[:, 1, 1] = cs
    else:
        for e, m in enumerate(meas):
            tm[e] = m.translation_vector
            Rm[e] = m.rotation_matrix
    return bi, tj, kap, tau, tm, Rm


def assemble(data, relaxation: str = QCQP_RELAXATION) -> ScoreModel:
    check_valid_relaxation(relaxation)
    d = data.dimension
    check_dimension(d)
    D1 = d + 1
    PB = d * D1  # columns per pose

    # ---- variables (gurobi_utils.py:221-310) ------------------------------
    pose_names = [p.name for chain in data.pose_variables for p in chain]
    landmark_names = [l.name for l in data.landmark_variables]
    _check_unique(pose_names, "pose_vars")
    _check_unique(landmark_names, "landmark_vars")
    for nm in landmark_names:
        if nm in set(pose_names):
            raise ValueError(f"Variable name {nm} already exists in pose_vars")
    range_keys = [(m.first_key, m.second_key) for m in data.range_measurements]
    if len(set(range_keys)) != len(range_keys):
        seen = set()
        for k in range_keys:
            if k in seen:
                raise ValueError(f"Variable name {k} already exists in distance_vars")
            seen.add(k)
    Np, Nl, Nr = len(pose_names), len(landmark_names), len(range_keys)
    if Np == 0:
        raise ValueError("factor graph has no pose variables")
    pose_idx = {nm: i for i, nm in enumerate(pose_names)}
    lm_idx = {nm: i for i, nm in enumerate(landmark_names)}
    lm_base = Np * PB
    rng_base = lm_base + Nl * d
    rw = 1 if relaxation == SOCP_RELAXATION else d
    n_model = rng_base + Nr * rw

    # first translation column (k = 0) and stride to the next k of every variable that has a
    # translation (VariableCollection.get_translation_var, gurobi_utils.py:103-109: poses first)
    tmap = {nm: (lm_base + i * d, 1) for nm, i in lm_idx.items()}
    tmap.update((nm, (i * PB + d, D1)) for nm, i in pose_idx.items())

    def tcols(name: str):
        try:
            return tmap[name]
        except KeyError:
            raise ValueError(f"Variable name {name} not found") from None

    rows, cols, vals = [], [], []  # residual Jacobian J (COO)
    cvec, wvec = [], []
    nrow = 0

    def add_block(r, c, v):
        rows.append(np.asarray(r, dtype=np.int64).ravel())
        cols.append(np.asarray(c, dtype=np.int64).ravel())
        vals.append(np.asarray(v, dtype=np.float64).ravel())

    # ---- relative-pose costs: odometry then loop closures (:380-430, :504-526)
    meas = [m for chain in data.odom_measurements for m in chain]
    meas += list(data.loop_closure_measurements)
    ne = len(meas)
    if ne:
        bi, tj, kap, tau, tm, Rm = _pose_meas_arrays(meas, pose_idx, d)
        k_ar = np.arange(d)
        # translation rows: t_j[k] - t_i[k] - sum_l R_i[k,l] tm[l]   (:516)
        r_t = nrow + (np.arange(ne)[:, None] * d + k_ar[None, :])  # (ne, d)
        add_block(r_t, tj[:, None] * PB + k_ar * D1 + d, np.ones((ne, d)))
        add_block(r_t, bi[:, None] * PB + k_ar * D1 + d, -np.ones((ne, d)))
        for l in range(d):
            add_block(r_t, bi[:, None] * PB + k_ar * D1 + l, -np.repeat(tm[:, l : l + 1], d, axis=1))
        cvec.append(np.zeros(ne * d))
        wvec.append(np.repeat(kap, d))
        nrow += ne * d
        # rotation rows: R_j[k,c] - sum_l R_i[k,l] Rm[l,c]          (:523)
        kk, cc = np.meshgrid(k_ar, k_ar, indexing="ij")  # (d, d)
        r_r = nrow + (np.arange(ne)[:, None, None] * d * d + kk[None] * d + cc[None])
        add_block(r_r, tj[:, None, None] * PB + kk[None] * D1 + cc[None], np.ones((ne, d, d)))
        for l in range(d):
            v = -np.broadcast_to(Rm[:, l, :][:, None, :], (ne, d, d))  # Rm[l, c]
            add_block(r_r, bi[:, None, None] * PB + kk[None] * D1 + l + 0 * cc[None], v)
        cvec.append(np.zeros(ne * d * d))
        wvec.append(np.repeat(tau, d * d))
        nrow += ne * d * d

    # ---- range costs (:449-501) ------------------------------------------
    if Nr:
        ends = np.array([tcols(ka) + tcols(kb) for ka, kb in range_keys], dtype=np.int64).reshape(Nr, 4)
        ta, sa, tb, sb = ends[:, 0], ends[:, 1], ends[:, 2], ends[:, 3]
        dist = np.fromiter((m.dist for m in data.range_measurements), dtype=np.float64, count=Nr)
        wr = np.fromiter((m.precision for m in data.range_measurements), dtype=np.float64, count=Nr)
        if relaxation == SOCP_RELAXATION:
            # w (d_ij - dist)^2   (:487)
            r_d = nrow + np.arange(Nr)
            add_block(r_d, rng_base + np.arange(Nr), np.ones(Nr))
            cvec.append(dist.copy())
            wvec.append(wr)
            nrow += Nr
        else:
            # w || t_i - t_j - dist * r_ij ||^2   (:489-496)
            k_ar = np.arange(d)
            r_q = nrow + np.arange(Nr)[:, None] * d + k_ar[None, :]
            add_block(r_q, ta[:, None] + sa[:, None] * k_ar, np.ones((Nr, d)))
            add_block(r_q, tb[:, None] + sb[:, None] * k_ar, -np.ones((Nr, d)))
            add_block(r_q, rng_base + np.arange(Nr)[:, None] * d + k_ar, -np.repeat(dist[:, None], d, axis=1))
            cvec.append(np.zeros(Nr * d))
            wvec.append(np.repeat(wr, d))
            nrow += Nr * d

    # ---- landmark priors (:433-446) --------------------------------------
    for prior in data.landmark_priors:
        t0, st = tcols(prior.name)
        tv = np.asarray(prior.translation_vector, dtype=np.float64)
        add_block(nrow + np.arange(d), t0 + st * np.arange(d), np.ones(d))
        cvec.append(tv)
        wvec.append(np.full(d, float(prior.translation_precision)))
        nrow += d

    # ---- pin the first pose of the first chain to [I | 0] (:181-183, :316-333):
    # x = (x_free, x_fixed), so J x - c = J_free x_free - (c - J_fixed x_fixed) and the pinned
    # columns never enter a matrix product
    first = data.pose_variables[0][0].name
    p0 = pose_idx[first]
    fixed_cols = p0 * PB + np.arange(PB)
    fixed_vals = np.hstack([np.eye(d), np.zeros((d, 1))]).ravel()
    mask = np.ones(n_model, dtype=bool)
    mask[fixed_cols] = False
    # Solver-space column order: replica by replica, one replica per matrix row k of the poses [R | t]
    # (the model only couples the entries of one row k at a time, except through the cones:
    # gurobi_utils.py:504-526, :345-352).  Replica k = [R(k,:) t(k)] of all free poses, chain by chain
    # (each block-tridiagonal chain of the preconditioner -- one per robot and row k -- owns a CONTIGUOUS
    # range of unknowns), then coordinate k of every landmark, then (QCQP) component k of every range
    # vector; the SOCP range variables form the tail.  P = I_d (x) P_row (+ tail) in this order: the
    # structure the solver's row-replication hint announces.  (Model space keeps the Gurobi layout;
    # expand()/reduce() map.)
    pieces_free = []
    j_ar = np.arange(D1)
    for k in range(d):
        base = 0
        for chain in data.pose_variables:
            L = len(chain)
            idx = base + np.arange(L)
            idx = idx[idx != p0]
            pieces_free.append((idx[:, None] * PB + k * D1 + j_ar[None, :]).ravel())
            base += L
        pieces_free.append(lm_base + np.arange(Nl) * d + k)
        if relaxation != SOCP_RELAXATION:
            pieces_free.append(rng_base + np.arange(Nr) * d + k)
    if relaxation == SOCP_RELAXATION:
        pieces_free.append(rng_base + np.arange(Nr))
    free_cols = np.concatenate(pieces_free)
    n_free = free_cols.size
    assert n_free == n_model - PB
    xc = np.zeros(n_model)
    xc[fixed_cols] = fixed_vals
    new_of_model = -np.ones(n_model, dtype=np.int64)
    new_of_model[free_cols] = np.arange(n_free)

    def reduced(r, cidx, v, nrows, const):
        """CSR on the free columns; the pinned part goes into `const -= M_fixed x_fixed`."""
        pinned = ~mask[cidx]
        if pinned.any():
            const = const - np.bincount(r[pinned], weights=v[pinned] * xc[cidx[pinned]], minlength=nrows)
            keep = ~pinned
            r, cidx, v = r[keep], cidx[keep], v[keep]
        M = sp.csr_matrix((v, (r, new_of_model[cidx])), shape=(nrows, n_free))
        return M, const

    if nrow:
        c = np.concatenate(cvec)
        w = np.concatenate(wvec)
        J, c = reduced(np.concatenate(rows), np.concatenate(cols), np.concatenate(vals), nrow, c)
        WJ = J.copy()
        WJ.data *= np.repeat(w, np.diff(J.indptr))
        Jt = J.T.tocsr()
        P = (Jt @ WJ).tocsr()
        P.data *= 2.0
        q = -2.0 * (Jt @ (w * c))
        c0 = _dot(c, w * c)
    else:
        P = sp.csr_matrix((n_free, n_free))
        q = np.zeros(n_free)
        c0 = 0.0

    # ---- cones (:336-352) --------------------------------------------------
    m = Nr * D1
    if Nr:
        k_ar = np.arange(d)
        if relaxation == SOCP_RELAXATION:
            # s = (d_ij, t_i - t_j) = b - A x  ->  A = -[e_d ; e_ti - e_tj], b = 0
            ar = [np.arange(Nr) * D1]
            ac = [rng_base + np.arange(Nr)]
            av = [-np.ones(Nr)]
            rr = (np.arange(Nr)[:, None] * D1 + 1 + k_ar[None, :])
            ar += [rr.ravel(), rr.ravel()]
            ac += [(ta[:, None] + sa[:, None] * k_ar).ravel(), (tb[:, None] + sb[:, None] * k_ar).ravel()]
            av += [-np.ones(Nr * d), np.ones(Nr * d)]
            b = np.zeros(m)
        else:
            # s = (1, r_ij)  ->  A = -[0 ; I], b = (1, 0..0)
            rr = (np.arange(Nr)[:, None] * D1 + 1 + k_ar[None, :])
            ar = [rr.ravel()]
            ac = [(rng_base + np.arange(Nr)[:, None] * d + k_ar).ravel()]
            av = [-np.ones(Nr * d)]
            b = np.zeros(m)
            b[np.arange(Nr) * D1] = 1.0
        A, b = reduced(np.concatenate(ar), np.concatenate(ac), np.concatenate(av), m, b)
    else:
        A = sp.csr_matrix((0, n_free))
        b = np.zeros(0)
    P.sum_duplicates(); P.sort_indices()
    A.sum_duplicates(); A.sort_indices()

    # ---- block-tridiagonal hint: one chain per (matrix row k, robot chain), replica by replica ----
    chain_ptr = [0]
    node_cols = []
    j_ar = np.arange(D1)
    for k in range(d):
        base = 0
        for chain in data.pose_variables:
            L = len(chain)
            idx = base + np.arange(L)
            # the pinned pose is not an unknown: a chain that contains it is cut there
            pieces = [idx] if not (base <= p0 < base + L) else [idx[idx < p0], idx[idx > p0]]
            for piece in pieces:
                if piece.size == 0:
                    continue
                node_cols.append(new_of_model[piece[:, None] * PB + k * D1 + j_ar[None, :]].ravel())
                chain_ptr.append(chain_ptr[-1] + piece.size)
            base += L
    n_rep = (Np - 1) * D1 + Nl + (0 if relaxation == SOCP_RELAXATION else Nr)
    qp = ConicQP(
        P=P, q=np.ascontiguousarray(q), c0=c0, A=A, b=np.ascontiguousarray(b), z=0,
        soc_dims=np.full(Nr, D1, dtype=np.int32),
        chain_ptr=np.asarray(chain_ptr, dtype=np.int32),
        node_cols=(np.concatenate(node_cols).astype(np.int32) if node_cols else np.zeros(0, np.int32)),
        block_size=D1, rep_d=d, rep_n=n_rep,
    )
    return ScoreModel(
        dim=d, relaxation=relaxation, qp=qp, n_model=n_model, free_cols=free_cols,
        fixed_cols=fixed_cols, fixed_vals=fixed_vals, pose_names=pose_names,
        landmark_names=landmark_names, range_keys=range_keys, lm_base=lm_base,
        rng_base=rng_base, rng_width=rw,
        range_ends=(ends if Nr else None), range_dist=(dist if Nr else None),
    )
