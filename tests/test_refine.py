"""Local refinement after SCORE (reference README.md:63-67): Gauss-Newton /
Levenberg-Marquardt on SE(2) from the SCORE estimate, its normal equations solved through the C ABI's
linear mode (here: the CPU twin's implementation of it; the HIP one in tests/test_gpu_parity.py), checked
against SciPy's least_squares run on the same residual function (an independent trust-region solver with a
finite-difference Jacobian check) and against the sparse-LU variant of the same loop."""
import numpy as np
import pytest
from scipy.optimize import least_squares

from score_amd.manhattan import make_manhattan
from score_amd.refine import _initial_point, _Problem, refine_estimate
from score_amd.solve_score import solve_score


def _graph():
    return make_manhattan(n_robots=2, n_poses=30, n_beacons=3, seed=77, p_range=0.5, sigma_t=0.05, sigma_theta=0.02)


def test_jacobian_matches_finite_differences(twin_lib):
    fg = _graph()
    res = solve_score(fg, "SOCP", lib_path=twin_lib)
    prob = _Problem(fg)
    u = _initial_point(prob, res) + 1e-2 * np.random.default_rng(0).standard_normal(prob.n)
    r, J = prob.residuals(u, jac=True)
    J = J.toarray()
    h = 1e-6
    for k in np.random.default_rng(1).choice(prob.n, size=25, replace=False):
        e = np.zeros(prob.n); e[k] = h
        fd = (prob.residuals(u + e) - prob.residuals(u - e)) / (2 * h)
        np.testing.assert_allclose(J[:, k], fd, atol=1e-5 * max(1.0, np.abs(fd).max()))


def test_refinement_reaches_the_least_squares_optimum(twin_lib):
    fg = _graph()
    res = solve_score(fg, "SOCP", lib_path=twin_lib)
    assert res.solved
    refined, info = refine_estimate(fg, res, lib_path=twin_lib)  # the whole loop behind the ABI (score_refine_run)
    assert info["engine"] == "native" and info["linear_solves"] >= 1 and info["pcg_iters"] >= 1
    # the Python loop (host Jacobians, normal equations through score_linear_solve) takes the same path:
    # same iterations, same PCG counts, same estimate
    by_py, info_py = refine_estimate(fg, res, lib_path=twin_lib, engine="python")
    assert (info_py["iterations"], info_py["linear_solves"]) == (info["iterations"], info["linear_solves"])
    assert abs(info_py["pcg_iters"] - info["pcg_iters"]) <= 2
    assert info_py["cost_final"] == pytest.approx(info["cost_final"], rel=1e-12)
    for nm in refined.poses:
        np.testing.assert_allclose(refined.poses[nm], by_py.poses[nm], atol=1e-9)
    by_lu, info_lu = refine_estimate(fg, res, linear_solver="scipy")
    assert info["cost_final"] == pytest.approx(info_lu["cost_final"], rel=1e-9)
    for nm in refined.poses:
        np.testing.assert_allclose(refined.poses[nm], by_lu.poses[nm], atol=1e-5)
    assert info["cost_final"] <= info["cost_initial"] + 1e-12 and info["grad_inf"] < 1e-5 * max(1.0, info["cost_final"])
    prob = _Problem(fg)
    u0 = _initial_point(prob, res)
    ref = least_squares(prob.residuals, u0, jac=lambda u: prob.residuals(u, jac=True)[1], method="trf", xtol=1e-14, ftol=1e-14, gtol=1e-12)
    # both reach the same local minimum (least_squares stops on its step tolerance a little earlier)
    f_ref = float(ref.fun @ ref.fun)
    assert info["cost_final"] <= f_ref + 1e-9 and info["cost_final"] == pytest.approx(f_ref, rel=1e-5)
    # an independent solver started AT the refined point finds nothing to improve: it is a minimiser
    # (the valley is flat along weakly observed directions, so two solvers stopping on their own
    # tolerances may sit millimetres apart; the cost and stationarity are what is well defined)
    u_ref = prob.pack(*prob.split(_initial_point(prob, refined)))
    again = least_squares(prob.residuals, u_ref, jac=lambda u: prob.residuals(u, jac=True)[1], method="trf", xtol=1e-14, ftol=1e-14, gtol=1e-12)
    assert float(again.fun @ again.fun) >= info["cost_final"] * (1 - 1e-9)
    assert np.abs(again.x - u_ref).max() < 1e-4
    # the pinned pose stays where SCORE put it; rotations are proper
    first = fg.pose_variables[0][0].name
    np.testing.assert_allclose(refined.poses[first], res.poses[first], atol=1e-12)
    for T in refined.poses.values():
        assert np.linalg.det(T[:2, :2]) == pytest.approx(1.0, abs=1e-12)
    # the refined estimate is at least as close to the ground truth as SCORE's initial estimate
    def rmse(r):
        err = [np.linalg.norm(r.poses[p.name][:2, 2] - np.asarray(p.true_position)) for chain in fg.pose_variables for p in chain]
        return float(np.sqrt(np.mean(np.square(err))))
    assert rmse(refined) <= rmse(res) + 1e-9


def test_linear_mode_solves_the_normal_equations(twin_lib):
    """score_linear_create / score_linear_solve on the damped normal equations of a graph with loop closures
    and ranges: the solution equals SciPy's sparse direct solve; new values on the same handle; errors."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    from score_amd.refine import _DeviceNormalEquations
    from score_amd.solver import LinearSolver

    fg = make_manhattan(n_robots=3, n_poses=40, n_beacons=2, seed=5, p_range=0.4, n_loop_closures=4)
    res = solve_score(fg, "SOCP", lib_path=twin_lib)
    prob = _Problem(fg)
    u = _initial_point(prob, res)
    r, J = prob.residuals(u, jac=True)
    dev = _DeviceNormalEquations(prob, J, twin_lib, None)
    try:
        H = (J.T @ J).tocsr()
        g = J.T @ r
        for lam in (1e-6, 1e-2):
            step = dev.solve(H, lam, -g, 1e-11)
            ref = spla.spsolve((H + lam * sp.identity(prob.n)).tocsc(), -g)
            np.testing.assert_allclose(step, ref, atol=1e-8 * max(1.0, np.abs(ref).max()))
        assert dev.pcg_iters > 0
        v = dev.values(H)
        v[dev.diag] += 1e-3
        x, info = dev.solver.solve(v, np.zeros(prob.n))
        assert info["converged"] and info["iters"] == 0 and not np.any(x)
        x, info = dev.solver.solve(v, -g, rel_tol=1e-10, max_iters=3, residual=True)  # iteration cap is reported
        assert not info["converged"] and info["iters"] == 3 and info["rel_residual"] > 1e-10
        with pytest.raises(ValueError):
            dev.solver.solve(v[:-1], -g)
    finally:
        dev.close()
    # pattern checks of score_linear_create
    n = 6
    good = sp.identity(n, format="csr")
    with pytest.raises(RuntimeError, match="diagonal"):
        LinearSolver(sp.csr_matrix(([1.0], ([0], [1])), shape=(n, n)), [0, 2], [0, 3], 3, lib_path=twin_lib)
    ls = LinearSolver(good, [0, 2], [0, 3], 3, lib_path=twin_lib)
    x, info = ls.solve(np.full(n, 4.0), np.arange(1.0, n + 1))
    np.testing.assert_allclose(x, np.arange(1.0, n + 1) / 4.0, rtol=1e-12)
    ls.close()


def test_native_refinement_blocks_match_the_python_jacobian(twin_lib):
    """score_gn.hpp's per-measurement blocks against J'J / J'r of the Python Jacobian: one LM step of the native
    engine from a perturbed point equals the Python engine's (pinned endpoints, loop closures, priors)."""
    from score_amd import compat

    fg = make_manhattan(n_robots=2, n_poses=25, n_beacons=2, seed=9, p_range=0.5, n_loop_closures=3)
    fg.landmark_priors = [compat.LandmarkPrior2D(name=fg.landmark_variables[0].name, position=(1.0, -2.0), translation_precision=3.0)]
    res = _noisy_truth(fg)  # (an estimate to start from; no SCORE solve needed for this check)
    a, ia = refine_estimate(fg, res, lib_path=twin_lib, max_iters=1)
    b, ib = refine_estimate(fg, res, lib_path=twin_lib, max_iters=1, engine="python")
    assert ia["cost_initial"] == pytest.approx(ib["cost_initial"], rel=1e-13)
    assert ia["cost_final"] == pytest.approx(ib["cost_final"], rel=1e-7)  # (one PCG solve to 1e-9 on each side)
    for nm in a.poses:
        np.testing.assert_allclose(a.poses[nm], b.poses[nm], atol=1e-7)
    for nm in a.landmarks:
        np.testing.assert_allclose(a.landmarks[nm], b.landmarks[nm], atol=1e-7)


def _noisy_truth(fg, seed=0):
    from score_amd import compat

    rng = np.random.default_rng(seed)
    names = [p.name for ch in fg.pose_variables for p in ch]
    T = np.tile(np.eye(3), (len(names), 1, 1))
    i = 0
    for ch in fg.pose_variables:
        for p in ch:
            th = p.true_theta + 0.02 * rng.normal()
            T[i, :2, :2] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]
            T[i, :2, 2] = np.asarray(p.true_position) + 0.1 * rng.normal(size=2)
            i += 1
    lms = np.array([np.asarray(l.true_position) + 0.1 * rng.normal(size=2) for l in fg.landmark_variables]).reshape(-1, 2)
    vals = compat.VariableValues(2, compat.ArrayDict(names, T), compat.ArrayDict([l.name for l in fg.landmark_variables], lms), None)
    return compat.SolverResults(variables=vals, total_time=0.0, solved=True, pose_chain_names=fg.get_pose_chain_names(),
                                solver_cost=0.0, info={})


def check_refinement_edge_cases(lib):
    """No ranges at all; a two-pose graph; a single (pinned) pose with one landmark (no chain nodes: the
    preconditioner is Jacobi only); linear mode without a chain hint."""
    import scipy.sparse as sp

    from score_amd.solver import LinearSolver

    fg = make_manhattan(n_robots=2, n_poses=15, n_beacons=0, seed=1, p_range=0.0, n_loop_closures=2)
    res = _noisy_truth(fg)
    a, ia = refine_estimate(fg, res, lib_path=lib)
    b, ib = refine_estimate(fg, res, linear_solver="scipy")
    assert ia["cost_final"] == pytest.approx(ib["cost_final"], rel=1e-9) and ia["iterations"] == ib["iterations"]
    fg = make_manhattan(n_robots=1, n_poses=2, n_beacons=1, seed=2, p_range=1.0)
    a, ia = refine_estimate(fg, _noisy_truth(fg), lib_path=lib)
    assert ia["cost_final"] < 1e-18
    fg = make_manhattan(n_robots=1, n_poses=1, n_beacons=1, seed=3, p_range=1.0)
    assert len(fg.range_measurements) == 1
    a, ia = refine_estimate(fg, _noisy_truth(fg), lib_path=lib)
    assert ia["cost_final"] < 1e-18 and ia["linear_solves"] >= 1
    K = sp.csr_matrix(np.array([[4.0, 1, 0], [1, 3, 0], [0, 0, 2]]))
    ls = LinearSolver(K, [0], [], 0, lib_path=lib)
    x, info = ls.solve(K.data, np.array([1.0, 2, 3]), rel_tol=1e-12)
    ls.close()
    assert info["converged"]
    np.testing.assert_allclose(x, np.linalg.solve(K.toarray(), [1.0, 2, 3]), rtol=1e-10)


def test_refinement_edge_cases(twin_lib):
    check_refinement_edge_cases(twin_lib)


# ---------------------------------------------------------------------------------------------
# 3-D: SE(3)^N x R^(3 L) (the reference's model is dimension-generic, gurobi_utils.py:37-50)
# ---------------------------------------------------------------------------------------------
def _graph3():
    from score_amd import compat
    from score_amd.manhattan import make_manhattan_3d

    fg = make_manhattan_3d(n_robots=2, n_poses=25, n_beacons=3, seed=41, p_range=0.5, sigma_t=0.05, sigma_theta=0.02)
    fg.landmark_priors = [compat.LandmarkPrior3D(fg.landmark_variables[1].name, (1.0, 2.0, -1.0), 0.5)]
    return fg


def _noisy_truth3(fg, seed=0):
    from score_amd import compat
    from score_amd.refine import so3_exp

    rng = np.random.default_rng(seed)
    names = [p.name for ch in fg.pose_variables for p in ch]
    T = np.tile(np.eye(4), (len(names), 1, 1))
    for i, p in enumerate(q for ch in fg.pose_variables for q in ch):
        T[i, :3, :3] = p.rotation_matrix @ so3_exp(0.03 * rng.normal(size=3))
        T[i, :3, 3] = np.asarray(p.true_position) + 0.1 * rng.normal(size=3)
    lms = np.array([np.asarray(l.true_position) + 0.1 * rng.normal(size=3) for l in fg.landmark_variables]).reshape(-1, 3)
    vals = compat.VariableValues(3, compat.ArrayDict(names, T), compat.ArrayDict([l.name for l in fg.landmark_variables], lms), None)
    return compat.SolverResults(variables=vals, total_time=0.0, solved=True, pose_chain_names=fg.get_pose_chain_names(),
                                solver_cost=0.0, info={})


def test_3d_jacobian_matches_finite_differences():
    """The tangent-space Jacobian of _Problem3D (retraction R Exp(omega), t + v) against central differences."""
    from score_amd.refine import _Problem3D

    fg = _graph3()
    prob = _Problem3D(fg)
    state = prob.initial_state(_noisy_truth3(fg))
    r, J = prob.residuals(state, jac=True)
    J = J.toarray()
    h = 1e-6
    for k in np.random.default_rng(1).choice(prob.n, size=30, replace=False):
        e = np.zeros(prob.n); e[k] = h
        fd = (prob.residuals(prob.retract(state, e)) - prob.residuals(prob.retract(state, -e))) / (2 * h)
        np.testing.assert_allclose(J[:, k], fd, atol=1e-6 * max(1.0, np.abs(fd).max()))


def check_3d_refinement(lib):
    """SE(3) refinement behind score_refine_create / score_refine_run (score_gn.hpp: 12 x 12 relative-pose blocks,
    retraction kernel, two 3 x 3 chains per robot) against the Python loop with SciPy's sparse LU and against
    scipy.optimize.least_squares on the same residuals (finite-difference Jacobian in a chart around the start)."""
    from score_amd.refine import _Problem3D

    fg = _graph3()
    res = _noisy_truth3(fg)
    a, ia = refine_estimate(fg, res, lib_path=lib)
    assert ia["engine"] == "native" and ia["linear_solves"] >= 1 and ia["pcg_iters"] >= 1
    b, ib = refine_estimate(fg, res, linear_solver="scipy")            # host Jacobians + sparse LU: the test reference
    c, ic = refine_estimate(fg, res, lib_path=lib, engine="python")    # host Jacobians + device linear solves
    assert ia["cost_initial"] == pytest.approx(ib["cost_initial"], rel=1e-12)
    assert ia["cost_final"] == pytest.approx(ib["cost_final"], rel=1e-8) and ic["cost_final"] == pytest.approx(ib["cost_final"], rel=1e-8)
    assert ia["cost_final"] < 0.2 * ia["cost_initial"] and ia["grad_inf"] < 1e-5 * max(1.0, ia["cost_final"])
    for nm in a.poses:
        np.testing.assert_allclose(a.poses[nm], b.poses[nm], atol=1e-5)
        R = a.poses[nm][:3, :3]
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-12)
        assert np.linalg.det(R) == pytest.approx(1.0, abs=1e-12)
    for nm in a.landmarks:
        np.testing.assert_allclose(a.landmarks[nm], b.landmarks[nm], atol=1e-5)
    first = fg.pose_variables[0][0].name
    np.testing.assert_array_equal(a.poses[first], res.poses[first])  # the pinned pose stays
    # an independent trust-region solver in a chart around the start reaches the same minimum value
    prob = _Problem3D(fg)
    s0 = prob.initial_state(res)
    ref = least_squares(lambda d: prob.residuals(prob.retract(s0, d)), np.zeros(prob.n), method="trf", xtol=1e-14, ftol=1e-14, gtol=1e-12)
    f_ref = float(ref.fun @ ref.fun)
    assert ia["cost_final"] <= f_ref * (1 + 1e-6) and ia["cost_final"] == pytest.approx(f_ref, rel=1e-4)


def test_3d_refinement_reaches_the_least_squares_optimum(twin_lib):
    check_3d_refinement(twin_lib)


def test_3d_refinement_after_score(twin_lib):
    """SCORE then local refinement on a 3-D graph: the refined estimate is no further from the truth."""
    from score_amd.manhattan import make_manhattan_3d

    fg = make_manhattan_3d(n_robots=2, n_poses=30, n_beacons=3, seed=43, p_range=0.5, sigma_t=0.05, sigma_theta=0.02)
    res = solve_score(fg, "SOCP", lib_path=twin_lib)
    assert res.solved
    refined, info = refine_estimate(fg, res, lib_path=twin_lib)
    assert info["cost_final"] <= info["cost_initial"] + 1e-12

    def rmse(r):  # (the pinned robot: the other one is tied to it by ranges only and keeps a gauge freedom)
        err = [np.linalg.norm(r.poses[p.name][:3, 3] - np.asarray(p.true_position)) for p in fg.pose_variables[0]]
        return float(np.sqrt(np.mean(np.square(err))))
    assert rmse(refined) <= rmse(res) + 1e-6


def check_long_pcg_verdict(lib):
    """A PCG solve of several hundred iterations (a stiff 1-D chain without a chain hint: Jacobi only): the device gate's
    verdict -- r'M^-1 r of the RECURRED residual below tol^2 -- must agree with the true residual |rhs - K x| / |rhs|
    computed from the returned x (every 32nd product w = K p is recomputed directly so that the recurrence cannot drift)."""
    import scipy.sparse as sp

    from score_amd.solver import LinearSolver

    n = 600
    main = (2.0 + 1e-6) * (1.0 + 0.3 * np.sin(np.arange(n)) ** 2)  # (varying diagonal: Jacobi is not a multiple of I)
    off = -np.sqrt(main[:-1] * main[1:]) / (2.0 + 1e-6)  # D^(1/2) (Laplacian + shift) D^(1/2): SPD, condition ~ 1e5
    K = sp.diags([off, main, off], [-1, 0, 1], format="csr")
    rhs = np.cos(0.37 * np.arange(n)) + 0.1
    ls = LinearSolver(K, [0], [], 0, lib_path=lib)
    x, info = ls.solve(K.data, rhs, rel_tol=1e-10, max_iters=4000, residual=True)
    ls.close()
    assert info["converged"] and info["iters"] > 150, info
    assert info["rel_residual"] < 1e-8, info  # (M^-1-norm against 2-norm: a factor of cond(M)^(1/2) at most)
    ref = sp.linalg.spsolve(K.tocsc(), rhs)
    np.testing.assert_allclose(x, ref, atol=1e-7 * np.abs(ref).max())


def test_long_pcg_verdict_matches_the_true_residual(twin_lib):
    check_long_pcg_verdict(twin_lib)
