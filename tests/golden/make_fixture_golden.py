"""Builds the committed fixtures (run in the build container only):

* manhattan_fg.npz / goats_fg.npz -- the reference's two shipped DATA files
  (examples/manhattan/factor_graph.pickle, examples/goats_14_data/
  goats_14_6_2002_15_20.pkl) re-encoded as plain arrays (pickles of
  py_factor_graph classes cannot be loaded without that package);
* *_golden.npz -- optimum of the SCORE relaxation for each fixture and for
  three small synthetic graphs, computed by the oracle's semismooth Newton
  method (oracle/score_oracle.py) and cross-checked here against the CPU twin
  of the ADMM solver: objective, pose matrices [R|t], landmark positions and a
  mask of the landmarks that the optimum determines uniquely (a landmark all
  of whose range cones are slack can sit anywhere in a region, SURVEY.md
  hard part 6).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import score_oracle as so  # noqa: E402
from score_amd.assemble import assemble  # noqa: E402
from score_amd.io import load_pyfg_pickle, save_fg_npz  # noqa: E402
from score_amd.manhattan import make_manhattan  # noqa: E402
from score_amd.solver import ConicSolver  # noqa: E402

TWIN = os.path.join(ROOT, "oracle", "cpu_twin", "libscore_cpu.so")

SYNTH = {
    "synth_a": dict(n_robots=1, n_poses=60, n_beacons=2, seed=11),
    "synth_b": dict(n_robots=3, n_poses=50, n_beacons=3, seed=12, n_loop_closures=4),
    "synth_c": dict(n_robots=2, n_poses=120, n_beacons=0, seed=13, p_range=0.3),
}


def golden_for(name, fg):
    rp, u, info = so.newton_solve(fg, tol=1e-14, max_iter=300)
    vals = so.reduced_to_values(rp, u, "SOCP")
    mdl = assemble(fg, "SOCP")
    sol = ConicSolver(mdl.qp, dict(eps_abs=1e-9, eps_rel=1e-9, max_iters=30000), lib_path=TWIN).solve()[0]
    xm = mdl.expand(sol.x)
    P_admm = mdl.pose_blocks(xm)
    L_admm = mdl.landmark_block(xm)
    P_newton = np.stack([vals["poses"][n] for n in mdl.pose_names])
    L_newton = (np.stack([vals["landmarks"][n] for n in mdl.landmark_names])
                if mdl.landmark_names else np.zeros((0, fg.dimension)))
    scale = max(1.0, float(np.max(np.abs(P_newton))))
    pose_diff = float(np.max(np.abs(P_admm - P_newton))) / scale
    lm_diff = np.max(np.abs(L_admm - L_newton), axis=1) / scale if len(L_newton) else np.zeros(0)
    determined = lm_diff < 1e-6
    # a pose is "determined" when two unrelated solvers agree on it: robots that
    # no active range cone ties to the pinned robot keep a gauge freedom
    pose_err = np.max(np.abs(P_admm - P_newton).reshape(len(P_newton), -1), axis=1) / scale
    pose_determined = pose_err < 1e-6
    n_first_chain = len(fg.pose_variables[0])
    obj_n = so.LiteralModel(fg, "SOCP").direct_cost(vals)
    print(f"{name}: newton iters {info['iters']} |g| {info['grad_inf']:.2e} obj {obj_n:.10f} | admm status "
          f"{sol.info['status']} iters {sol.info['iters']} pobj {sol.info['pobj']:.10f} | pose diff {pose_diff:.2e} "
          f"landmark diff {lm_diff} determined {determined}")
    assert pose_determined[:n_first_chain].all(), "oracle and ADMM twin disagree on the pinned robot's poses"
    print(f"   determined poses: {int(pose_determined.sum())} of {len(pose_determined)}")
    assert abs(obj_n - sol.info["pobj"]) < 1e-5 * max(1.0, abs(obj_n))
    np.savez_compressed(
        os.path.join(HERE, f"{name}_golden.npz"),
        objective=np.float64(obj_n), poses=P_newton, landmarks=L_newton, landmark_determined=determined, pose_determined=pose_determined,
        pose_names=np.array(mdl.pose_names, dtype="U"), landmark_names=np.array(mdl.landmark_names, dtype="U"),
    )


if __name__ == "__main__":
    fixtures = {
        "manhattan": "/root/reference/examples/manhattan/factor_graph.pickle",
        "goats": "/root/reference/examples/goats_14_data/goats_14_6_2002_15_20.pkl",
    }
    for name, path in fixtures.items():
        fg = load_pyfg_pickle(path)
        save_fg_npz(os.path.join(HERE, f"{name}_fg.npz"), fg)
        golden_for(name, fg)
    for name, kw in SYNTH.items():
        golden_for(name, make_manhattan(**kw))
