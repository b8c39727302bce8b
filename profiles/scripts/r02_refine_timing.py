"""Local refinement after SCORE (f4) on the headline graph: normal equations on the GPU (score_linear_solve)
against SciPy sparse LU, same Levenberg-Marquardt loop."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import numpy as np
from score_amd.manhattan import make_manhattan
from score_amd.refine import refine_estimate, _DeviceNormalEquations, _Problem, _initial_point
from score_amd.solve_score import solve_score
for (r, n, b) in ((4, 1000, 4), (20, 1000, 4)):
    fg = make_manhattan(n_robots=r, n_poses=n, n_beacons=b, seed=3000)
    res = solve_score(fg, "SOCP")
    for which, eng in (("device", "native"), ("device", "native"), ("device", "python"), ("scipy", "python")):
        t = time.perf_counter(); out, info = refine_estimate(fg, res, linear_solver=which, engine=eng); dt = time.perf_counter() - t
        if eng == "native":
            print(f"     (native: create {info['setup_ms']:.1f} ms, run {info['solve_ms']:.1f} ms)")
        print(f"{r}x{n} {which:6s} {eng:6s}: {dt*1e3:8.1f} ms  LM its {info['iterations']}  cost {info['cost_initial']:.6f} -> {info['cost_final']:.6f}  "
              f"|g| {info['grad_inf']:.2e}  linear solves {info['linear_solves']}  PCG its {info['pcg_iters']}", flush=True)
    prob = _Problem(fg); u = _initial_point(prob, res); rr, J = prob.residuals(u, jac=True)
    dev = _DeviceNormalEquations(prob, J, None, None)
    H = (J.T @ J).tocsr(); g = J.T @ rr
    v = dev.values(H); v[dev.diag] += 1e-6
    for tol in (1e-6, 1e-9):
        dev.solver.solve(v, -g, rel_tol=tol)
        t = time.perf_counter(); x, info = dev.solver.solve(v, -g, rel_tol=tol, max_iters=4000, residual=True); dt = time.perf_counter() - t
        print(f"   one solve n={prob.n} nnz={v.size} tol {tol:g}: {dt*1e3:.2f} ms, {info}", flush=True)
    dev.close()
