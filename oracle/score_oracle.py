"""ORACLE -- test infrastructure only (never imported by the product path).

CPU restatement of the SCORE relaxation that the reference builds in
``score/utils/gurobi_utils.py`` and hands to Gurobi's barrier solver
(``score/solve_score.py:76``).  Only tests/, ``__graft_entry__.smoke()`` and
bench.py's ``cpu_baseline`` leg may import this package.

PARITY STATUS
  * SO(d) rounding (``round_to_special_orthogonal`` below) is PINNED: checked
    against golden vectors produced by importing the reference's own
    ``score/utils/matrix_utils.py`` in the build container
    (tests/golden/make_rounding_golden.py -> tests/golden/rounding_golden.npz).
  * The conic solve is PARITY UNPINNED by the reference: gurobipy (proprietary,
    unpinned in setup.cfg:6) and py_factor_graph are absent from the reference
    tree and from every machine of this project, and the reference ships no
    tests or golden outputs.  What stands in: (1) ``direct_cost`` /
    ``cone_violation`` below evaluate the reference's objective and
    constraints literally, measurement by measurement, from the [R|t]
    matrices (no sparse algebra shared with the product assembler);
    (2) ``newton_solve`` minimises the equivalent reduced problem
    (min over d_ij>=|D| of w(d-dist)^2 == w*max(0,|D|-dist)^2)
    with a semismooth Newton method -- an algorithm unrelated to the product's
    ADMM; (3) ``kkt_certificate`` verifies any primal/dual pair against the
    conic optimality conditions, whichever solver produced it.

Every function cites the reference lines it restates.  Pure-Python loops are
used on purpose (clarity over speed); sizes are those of the fixtures.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as la
import scipy.sparse as sp
import scipy.sparse.linalg as spla


# ---------------------------------------------------------------------------
# rounding  (score/utils/matrix_utils.py:59-79, :293-318; used at
# score/utils/gurobi_utils.py:115-125)
# ---------------------------------------------------------------------------
def check_rotation_matrix(R: np.ndarray) -> None:
    """matrix_utils.py:293-318 with assert_test=True."""
    d = R.shape[0]
    if not np.allclose(R @ R.T, np.eye(d), rtol=1e-3, atol=1e-3):
        raise ValueError(f"R is not orthogonal {R @ R.T}")
    if not abs(np.linalg.det(R) - 1) < 1e-3:
        raise ValueError(f"R det incorrect {np.linalg.det(R)}")


def round_to_special_orthogonal(mat: np.ndarray) -> np.ndarray:
    """matrix_utils.py:59-79: U V^T, last singular direction flipped if det < 0."""
    assert mat.shape[0] == mat.shape[1], "matrix must be square"
    dim = mat.shape[0]
    try:
        S, D, Vh = la.svd(mat)
        R_so = S @ Vh
        if np.linalg.det(R_so) < 0:
            R_so = S @ np.diag([1] * (dim - 1) + [-1]) @ Vh
        check_rotation_matrix(R_so)
    except ValueError:
        raise ValueError(f"Could not round matrix to special orthogonal form: {mat}")
    return R_so


def clean_pose_est(pose_est: np.ndarray, dim: int) -> np.ndarray:
    """gurobi_utils.py:115-125: homogeneous pose with rounded rotation."""
    out = np.eye(dim + 1)
    out[:dim, :dim] = round_to_special_orthogonal(pose_est[:dim, :dim])
    out[:dim, -1] = pose_est[:dim, -1]
    return out


# ---------------------------------------------------------------------------
# literal evaluation of the reference's model on a candidate solution
# ---------------------------------------------------------------------------
class LiteralModel:
    """Variables exactly as the reference declares them (gurobi_utils.py:221-310):
    per pose a d x (d+1) matrix, per landmark a d-vector, per range either a
    scalar (SOCP) or a d-vector (QCQP), addressed by NAME like
    ``VariableCollection`` (:53-136)."""

    def __init__(self, data, relaxation: str):
        if relaxation not in ("SOCP", "QCQP"):  # :139-144
            raise ValueError(f"Relaxation {relaxation} is not supported.")
        if data.dimension not in (2, 3):  # :37-50
            raise ValueError(f"Value {data.dimension} is not 2 or 3")
        self.data = data
        self.relaxation = relaxation
        self.dim = data.dimension
        self.pose_names = [p.name for chain in data.pose_variables for p in chain]
        self.landmark_names = [l.name for l in data.landmark_variables]
        self.range_keys = [(m.first_key, m.second_key) for m in data.range_measurements]
        self.first_pose = data.pose_variables[0][0].name  # :181

    # -- value containers ---------------------------------------------------
    def pack(self, poses: dict, landmarks: dict, dists: dict) -> dict:
        return {"poses": poses, "landmarks": landmarks, "dists": dists}

    def translation(self, vals, name):  # :103-109
        if name in vals["poses"]:
            return vals["poses"][name][:, -1]
        if name in vals["landmarks"]:
            return vals["landmarks"][name]
        raise ValueError(f"Variable name {name} not found")

    # -- objective (:358-526) -------------------------------------------------
    def rel_pose_cost(self, Xi, Xj, meas) -> float:  # :504-526
        t_i, t_j = Xi[:, -1], Xj[:, -1]
        R_i, R_j = Xi[:, :-1], Xj[:, :-1]
        term = t_j - t_i - R_i @ np.asarray(meas.translation_vector)
        trans_obj = meas.translation_precision * float(term @ term)
        diff = R_j - R_i @ np.asarray(meas.rotation_matrix)
        rot_obj = meas.rotation_precision * float(np.sum(diff * diff))
        return rot_obj + trans_obj

    def direct_cost(self, vals) -> float:
        d = self.data
        cost = 0.0
        for chain in d.odom_measurements:  # :380-404
            for m in chain:
                cost += self.rel_pose_cost(vals["poses"][m.base_pose], vals["poses"][m.to_pose], m)
        for m in d.loop_closure_measurements:  # :407-430
            cost += self.rel_pose_cost(vals["poses"][m.base_pose], vals["poses"][m.to_pose], m)
        for m in d.range_measurements:  # :449-501
            key = (m.first_key, m.second_key)
            t_i = self.translation(vals, key[0])
            t_j = self.translation(vals, key[1])
            dv = vals["dists"][key]
            if self.relaxation == "SOCP":
                dij = float(np.asarray(dv).reshape(-1)[0])
                unweighted = m.dist ** 2 - 2 * m.dist * dij + dij ** 2  # :487
            else:
                inter = t_i - t_j - np.asarray(dv) * m.dist  # :489
                unweighted = float(inter @ inter)
            cost += m.precision * unweighted  # :500
        for pr in d.landmark_priors:  # :433-446
            t = self.translation(vals, pr.name) - np.asarray(pr.translation_vector)
            cost += pr.translation_precision * float(t @ t)
        return cost

    # -- constraints (:316-352) ----------------------------------------------
    def pin_violation(self, vals) -> float:
        X = vals["poses"][self.first_pose]
        d = self.dim
        return float(np.max(np.abs(X - np.hstack([np.eye(d), np.zeros((d, 1))]))))

    def cone_violation(self, vals) -> float:
        """max over ranges of the (positive part of the) constraint residual:
        SOCP |t_i - t_j| - d_ij and -d_ij (lb = 0, :290-293); QCQP |r_ij| - 1."""
        worst = 0.0
        for key in self.range_keys:
            dv = np.asarray(vals["dists"][key]).reshape(-1)
            if self.relaxation == "SOCP":
                diff = self.translation(vals, key[0]) - self.translation(vals, key[1])
                worst = max(worst, float(np.linalg.norm(diff) - dv[0]), float(-dv[0]))
            else:
                worst = max(worst, float(np.linalg.norm(dv) - 1.0))
        return worst


# ---------------------------------------------------------------------------
# reduced problem + semismooth Newton
# ---------------------------------------------------------------------------
class ReducedProblem:
    """Unknowns: all pose matrices except the pinned one, and the landmarks.
    F(u) = sum_quadratic_rows w (J u - c)^2 + sum_ranges w max(0, |D u + e| - dist)^2.
    The Jacobian rows are produced measurement by measurement with Python
    loops straight from gurobi_utils.py:504-526 / :433-446."""

    def __init__(self, data):
        self.data = data
        d = self.dim = data.dimension
        self.pose_names = [p.name for chain in data.pose_variables for p in chain]
        self.landmark_names = [l.name for l in data.landmark_variables]
        self.first_pose = data.pose_variables[0][0].name
        self.col = {}
        k = 0
        for nm in self.pose_names:
            if nm == self.first_pose:
                continue
            self.col[nm] = k
            k += d * (d + 1)
        for nm in self.landmark_names:
            self.col[nm] = k
            k += d
        self.n = k
        self.pin = np.hstack([np.eye(d), np.zeros((d, 1))])
        self._build_quadratic()
        self._build_ranges()

    # entry (k, j) of pose `nm`: returns (column or None, constant value)
    def _pe(self, nm, k, j):
        if nm == self.first_pose:
            return None, self.pin[k, j]
        return self.col[nm] + k * (self.dim + 1) + j, 0.0

    def _te(self, nm, k):
        if nm in self.col and nm in self._pose_set:
            return self._pe(nm, k, self.dim)
        if nm == self.first_pose:
            return None, 0.0
        return self.col[nm] + k, 0.0  # landmark

    def _build_quadratic(self):
        d = self.dim
        self._pose_set = set(self.pose_names)
        rows, cols, vals, cst, wts = [], [], [], [], []
        r = 0

        def emit(terms, w):
            nonlocal r
            const = 0.0
            for (c, cv), coef in terms:
                if c is None:
                    const += coef * cv
                else:
                    rows.append(r); cols.append(c); vals.append(coef)
            cst.append(-const)  # residual = J u - cst
            wts.append(w)
            r += 1

        meas = [m for chain in self.data.odom_measurements for m in chain]
        meas += list(self.data.loop_closure_measurements)
        for m in meas:
            tm = np.asarray(m.translation_vector, dtype=float)
            Rm = np.asarray(m.rotation_matrix, dtype=float)
            i, j = m.base_pose, m.to_pose
            for k in range(d):  # t_j - t_i - R_i tm     (:516)
                terms = [(self._pe(j, k, d), 1.0), (self._pe(i, k, d), -1.0)]
                terms += [(self._pe(i, k, l), -tm[l]) for l in range(d)]
                emit(terms, m.translation_precision)
            for k in range(d):  # R_j - R_i Rm          (:523)
                for c in range(d):
                    terms = [(self._pe(j, k, c), 1.0)]
                    terms += [(self._pe(i, k, l), -Rm[l, c]) for l in range(d)]
                    emit(terms, m.rotation_precision)
        for pr in self.data.landmark_priors:  # :433-446
            tv = np.asarray(pr.translation_vector, dtype=float)
            for k in range(d):
                c, cv = self._te(pr.name, k)
                rows.append(r); cols.append(c); vals.append(1.0)
                cst.append(tv[k]); wts.append(pr.translation_precision); r += 1
        self.J = sp.csr_matrix((vals, (rows, cols)), shape=(r, self.n))
        self.c = np.asarray(cst)
        self.w = np.asarray(wts)
        JW = self.J.T.multiply(self.w).tocsr()
        self.H0 = (2.0 * JW @ self.J).tocsc()
        self.g0 = -2.0 * (JW @ self.c)
        self.f0 = float(self.c @ (self.w * self.c))

    def _build_ranges(self):
        d = self.dim
        ms = self.data.range_measurements
        self.nr = len(ms)
        self.dist = np.array([m.dist for m in ms], dtype=float)
        self.wr = np.array([m.precision for m in ms], dtype=float)
        rows, cols, vals = [], [], []
        e = np.zeros(self.nr * d)
        for r, m in enumerate(ms):
            for k in range(d):
                for nm, sgn in ((m.first_key, 1.0), (m.second_key, -1.0)):
                    c, cv = self._te(nm, k)
                    if c is None:
                        e[r * d + k] += sgn * cv
                    else:
                        rows.append(r * d + k); cols.append(c); vals.append(sgn)
        self.D = sp.csr_matrix((vals, (rows, cols)), shape=(self.nr * d, self.n))
        self.e = e

    def deltas(self, u):
        return (self.D @ u + self.e).reshape(self.nr, self.dim)

    def value(self, u):
        # sum of squares, evaluated residual by residual (no cancellation)
        res = self.J @ u - self.c
        f = float(np.sum(self.w * res * res))
        if self.nr:
            rho = np.linalg.norm(self.deltas(u), axis=1)
            ex = np.maximum(0.0, rho - self.dist)
            f += float(np.sum(self.wr * ex * ex))
        return float(f)

    def grad_hess(self, u):
        d = self.dim
        g = self.H0 @ u + self.g0
        H = self.H0
        if self.nr:
            dl = self.deltas(u)
            rho = np.linalg.norm(dl, axis=1)
            act = rho > self.dist
            ex = np.where(act, rho - self.dist, 0.0)
            uh = np.zeros_like(dl)
            nz = rho > 0
            uh[nz] = dl[nz] / rho[nz, None]
            g = g + self.D.T @ (2.0 * (self.wr * ex)[:, None] * uh).ravel()
            # generalised Hessian block per active range:
            # 2w [ (1 - dist/rho)(I - uu') + uu' ]
            blocks = np.zeros((self.nr, d, d))
            a = np.where(act & nz, 1.0 - np.where(nz, self.dist / np.where(nz, rho, 1.0), 0.0), 0.0)
            eye = np.eye(d)[None]
            uu = uh[:, :, None] * uh[:, None, :]
            blocks = 2.0 * self.wr[:, None, None] * np.where(act[:, None, None], a[:, None, None] * (eye - uu) + uu, 0.0)
            B = sp.block_diag(list(blocks), format="csr") if self.nr else None
            H = (self.H0 + self.D.T @ B @ self.D).tocsc()
        return g, H


def newton_solve(data, tol: float = 1e-11, max_iter: int = 200, verbose: bool = False):
    """Semismooth Newton with backtracking on the reduced problem.  Returns
    (ReducedProblem, u*, info).  Stops when |grad|_inf <= tol * max(1, |g0|_inf)
    or when neither the objective nor the gradient norm can be improved."""
    rp = ReducedProblem(data)
    u = np.zeros(rp.n)
    d = rp.dim
    for nm in rp.pose_names:  # rotations = identity so the chains are not degenerate
        if nm == rp.first_pose:
            continue
        c = rp.col[nm]
        for k in range(d):
            u[c + k * (d + 1) + k] = 1.0
    f = rp.value(u)
    reg = 1e-10
    info = {"iters": 0}
    scale = max(1.0, float(np.max(np.abs(rp.g0))) if rp.g0.size else 1.0)
    g, H = rp.grad_hess(u)
    gn = float(np.max(np.abs(g))) if g.size else 0.0
    stall = 0
    for it in range(max_iter):
        if verbose:
            print(f"  newton it {it:3d} f={f:.12g} |g|inf={gn:.3e}")
        if gn <= tol * scale:
            break
        Hr = (H + reg * sp.identity(rp.n, format="csc")).tocsc()
        step = -spla.splu(Hr).solve(g)
        gs = float(g @ step)
        t = 1.0
        accepted = False
        while t >= 1e-10:
            un = u + t * step
            fn = rp.value(un)
            # near the optimum the decrease drowns in rounding: accept on the gradient instead
            if fn <= f + 1e-4 * t * gs or abs(t * gs) <= 1e-13 * max(1.0, abs(f)):
                gnew, Hnew = rp.grad_hess(un)
                gnn = float(np.max(np.abs(gnew)))
                if fn <= f + 1e-4 * t * gs or gnn < gn:
                    accepted = True
                    break
            t *= 0.5
        if not accepted:
            break
        tiny_decrease = (f - fn) <= 1e-12 * max(1.0, abs(f))
        stall = stall + 1 if (gnn >= 0.5 * gn and tiny_decrease) else 0
        u, f, g, H, gn = un, fn, gnew, Hnew, gnn
        info["iters"] = it + 1
        if stall >= 8:
            break
    info["grad_inf"] = gn
    info["objective"] = f
    return rp, u, info


def gauge_basis(rp: ReducedProblem) -> np.ndarray:
    """A basis (n x k, dense) of the null space of the ODOMETRY part of the objective, which
    contains the null space of the whole generalised Hessian: the directions along which the
    relative-pose costs (gurobi_utils.py:504-526) do not change.  For every pose chain whose first
    pose is free, perturbing that pose by (dR_0, dt_0) and propagating dR_{i+1} = dR_i Rm_i,
    dt_{i+1} = dt_i + dR_i tm_i along the chain leaves every odometry residual unchanged
    (d^2 + d directions per chain; none for the chain that holds the pinned pose, and the
    remainder of a chain cut by the pin starts at a fixed pose).  Every landmark coordinate is a
    direction of its own (landmarks appear in no odometry term)."""
    d = rp.dim
    cols = []
    for chain, odo in zip(rp.data.pose_variables, rp.data.odom_measurements):
        names = [p.name for p in chain]
        if rp.first_pose in names and names[0] == rp.first_pose:
            continue  # rooted at the pinned pose
        if rp.first_pose in names:
            raise NotImplementedError("pinned pose in the interior of a chain")
        by_edge = {(m.base_pose, m.to_pose): m for m in odo}
        for a in range(d):
            for b in range(d + 1):  # unit perturbation of entry (a, b) of [R_0 | t_0]
                v = np.zeros(rp.n)
                dR = np.zeros((d, d)); dt = np.zeros(d)
                if b < d:
                    dR[a, b] = 1.0
                else:
                    dt[a] = 1.0
                for i, nm in enumerate(names):
                    c = rp.col[nm]
                    blk = np.hstack([dR, dt[:, None]])
                    v[c : c + d * (d + 1)] = blk.ravel()
                    if i + 1 < len(names):
                        m = by_edge[(nm, names[i + 1])]
                        dt = dt + dR @ np.asarray(m.translation_vector, dtype=float)
                        dR = dR @ np.asarray(m.rotation_matrix, dtype=float)
                cols.append(v)
    for nm in rp.landmark_names:
        for k in range(d):
            v = np.zeros(rp.n)
            v[rp.col[nm] + k] = 1.0
            cols.append(v)
    return np.stack(cols, axis=1) if cols else np.zeros((rp.n, 0))


def determined_masks(rp: ReducedProblem, u: np.ndarray, rel_tol: float = 1e-9, active_tol: float = 1e-8):
    """Which poses / landmarks does the optimum determine uniquely?  Computed from the oracle's own
    objects only (no second solver).  F is convex and piecewise quadratic, so near an optimum u the
    optimal set is u + {v : H v = 0} with H the generalised Hessian at u (cones strictly active /
    strictly slack).  null(H) lies inside the span of gauge_basis(); restricted to that small basis
    N the test is an eigen-decomposition of N'HN.  A variable is determined iff its entries vanish in
    every null direction.  Returns (pose_mask, landmark_mask, info)."""
    d = rp.dim
    # generalised Hessian with STRICTLY active cones only: a cone whose excess is zero to rounding
    # (the iterate sits on the boundary of a slack region) does not pin anything -- the variable may
    # move inwards at no cost
    H = rp.H0
    if rp.nr:
        dl = rp.deltas(u)
        rho = np.linalg.norm(dl, axis=1)
        act = (rho - rp.dist) > active_tol * max(1.0, float(np.abs(u).max()))
        uh = np.zeros_like(dl)
        nz = rho > 0
        uh[nz] = dl[nz] / rho[nz, None]
        a = np.where(act & nz, 1.0 - rp.dist / np.where(nz, rho, 1.0), 0.0)
        uu = uh[:, :, None] * uh[:, None, :]
        blocks = 2.0 * rp.wr[:, None, None] * np.where(act[:, None, None], a[:, None, None] * (np.eye(d)[None] - uu) + uu, 0.0)
        B = sp.block_diag(list(blocks), format="csr")
        H = (rp.H0 + rp.D.T @ B @ rp.D).tocsc()
    N = gauge_basis(rp)
    if N.shape[1] == 0:
        return np.ones(len(rp.pose_names), bool), np.ones(len(rp.landmark_names), bool), {"null_dim": 0, "basis": 0}
    # scale the basis columns: propagated translations grow along a chain
    N = N / np.maximum(1e-300, np.abs(N).max(axis=0))
    G = N.T @ (H @ N)
    G = 0.5 * (G + G.T)
    w, V = np.linalg.eigh(G)
    top = max(float(w.max()), 1e-300)
    null = V[:, w <= rel_tol * top]
    Z = N @ null  # n x null_dim
    zmax = np.abs(Z).max() if Z.size else 0.0
    pose_mask = np.ones(len(rp.pose_names), bool)
    for i, nm in enumerate(rp.pose_names):
        if nm == rp.first_pose or Z.shape[1] == 0:
            continue
        c = rp.col[nm]
        pose_mask[i] = np.abs(Z[c : c + d * (d + 1)]).max() <= 1e-7 * max(1.0, zmax)
    lm_mask = np.ones(len(rp.landmark_names), bool)
    for i, nm in enumerate(rp.landmark_names):
        if Z.shape[1] == 0:
            continue
        c = rp.col[nm]
        lm_mask[i] = np.abs(Z[c : c + d]).max() <= 1e-7 * max(1.0, zmax)
    return pose_mask, lm_mask, {"null_dim": int(Z.shape[1]), "basis": int(N.shape[1]),
                                "eigs": w.tolist()}


def optimal_residuals(rp: ReducedProblem, u: np.ndarray):
    """Quantities every optimum shares even where the poses themselves are not unique: F is a
    strictly convex function of the residual vector, so the weighted relative-pose residuals J u - c
    and the range excesses max(0, |D u + e| - dist) are the same at every minimiser."""
    res = rp.J @ u - rp.c
    if rp.nr:
        ex = np.maximum(0.0, np.linalg.norm(rp.deltas(u), axis=1) - rp.dist)
    else:
        ex = np.zeros(0)
    return res, ex


def reduced_to_values(rp: ReducedProblem, u: np.ndarray, relaxation: str):
    """Expand the reduced optimum to the reference's variables: SOCP
    d_ij = max(|D|, dist); QCQP r_ij = D / max(|D|, dist)  (SURVEY.md 3.3)."""
    d = rp.dim
    poses, landmarks, dists = {}, {}, {}
    for nm in rp.pose_names:
        if nm == rp.first_pose:
            poses[nm] = rp.pin.copy()
        else:
            c = rp.col[nm]
            poses[nm] = u[c : c + d * (d + 1)].reshape(d, d + 1).copy()
    for nm in rp.landmark_names:
        c = rp.col[nm]
        landmarks[nm] = u[c : c + d].copy()
    dl = rp.deltas(u) if rp.nr else np.zeros((0, d))
    rho = np.linalg.norm(dl, axis=1)
    for r, m in enumerate(rp.data.range_measurements):
        key = (m.first_key, m.second_key)
        if relaxation == "SOCP":
            dists[key] = np.array([max(rho[r], rp.dist[r])])
        else:
            den = max(rho[r], rp.dist[r])
            dists[key] = dl[r] / den if den > 0 else np.zeros(d)
    return {"poses": poses, "landmarks": landmarks, "dists": dists}


# ---------------------------------------------------------------------------
# solver-independent optimality certificate for the conic form
# ---------------------------------------------------------------------------
def proj_soc(v: np.ndarray) -> np.ndarray:
    t, z = v[0], v[1:]
    nz = np.linalg.norm(z)
    if nz <= t:
        return v.copy()
    if nz <= -t:
        return np.zeros_like(v)
    a = 0.5 * (t + nz)
    out = np.empty_like(v)
    out[0] = a
    out[1:] = a * z / nz
    return out


def proj_cone(v: np.ndarray, z: int, soc_dims) -> np.ndarray:
    out = np.empty_like(v)
    out[:z] = 0.0
    k = z
    for dm in soc_dims:
        out[k : k + dm] = proj_soc(v[k : k + dm])
        k += dm
    return out


def kkt_certificate(P, q, A, b, z, soc_dims, x, y, s=None):
    """Residuals of: Ax + s = b, s in K, y in K* (zero-cone duals free),
    Px + q + A'y = 0, s'y = 0; plus the duality gap."""
    if s is None:
        s = b - A @ x
    pri = b - A @ x - s
    s_dist = s - proj_cone(s, z, soc_dims)
    yk = y.copy()
    yk[:z] = 0.0
    yproj = proj_cone(yk, 0, np.concatenate([[1] * 0, soc_dims]).astype(int)) if z == 0 else None
    if z:
        yy = y[z:]
        y_dist = yy - proj_cone(yy, 0, soc_dims)
    else:
        y_dist = y - yproj
    dual = P @ x + q + A.T @ y
    pobj = 0.5 * x @ (P @ x) + q @ x
    dobj = -0.5 * x @ (P @ x) - b @ y
    return {
        "primal_res_inf": float(np.max(np.abs(pri))) if pri.size else 0.0,
        "s_cone_dist": float(np.max(np.abs(s_dist))) if s.size else 0.0,
        "y_cone_dist": float(np.max(np.abs(y_dist))) if y_dist.size else 0.0,
        "dual_res_inf": float(np.max(np.abs(dual))) if dual.size else 0.0,
        "complementarity": float(abs(s @ y)),
        "gap": float(abs(pobj - dobj)),
        "pobj": float(pobj),
        "dobj": float(dobj),
    }
