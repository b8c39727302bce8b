"""Local refinement after SCORE (reference README.md:63-67), on the CPU: Gauss-Newton /
Levenberg-Marquardt on SE(2) from the SCORE estimate, checked against SciPy's least_squares run on the
same residual function (an independent trust-region solver with a finite-difference Jacobian check)."""
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
    refined, info = refine_estimate(fg, res)
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
