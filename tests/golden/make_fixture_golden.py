"""Builds the committed fixtures (run in the build container only):

* manhattan_fg.npz / goats_fg.npz -- the reference's two shipped DATA files
  (examples/manhattan/factor_graph.pickle, examples/goats_14_data/
  goats_14_6_2002_15_20.pkl) re-encoded as plain arrays (pickles of
  py_factor_graph classes cannot be loaded without that package);
* *_golden.npz -- optimum of the SCORE relaxation for each fixture, for four small
  synthetic 2-D graphs and for one 3-D graph, computed by the oracle's semismooth
  Newton method (oracle/score_oracle.py).  Everything in a golden file comes from
  the ORACLE ALONE -- the product's assembler, setup code and solvers are not
  involved (round 1 derived the masks from agreement with the CPU twin, which
  shares code with the product):
    objective, pose matrices [R|t], landmark positions;
    pose_determined / landmark_determined -- which variables the optimum fixes
      uniquely, from the null space of the generalised Hessian at the optimum
      (oracle.determined_masks: robots no active cone ties to the pinned robot
      keep a gauge freedom, a landmark all of whose cones are slack can move);
    quad_residuals, range_excess -- the relative-pose residuals and the range
      excesses max(0, |t_i - t_j| - d_ij), which EVERY optimum shares (F is
      strictly convex in them), so that solutions can also be compared where the
      poses themselves are not unique.
  The CPU twin is still run, but only to print a sanity line.
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

sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import SYNTH, graph_3d, graph_by_name  # noqa: E402


def golden_for(name, fg):
    rp, u, info = so.newton_solve(fg, tol=1e-14, max_iter=300)
    vals = so.reduced_to_values(rp, u, "SOCP")
    pose_names = [p.name for chain in fg.pose_variables for p in chain]
    landmark_names = [l.name for l in fg.landmark_variables]
    P_newton = np.stack([vals["poses"][n] for n in pose_names])
    L_newton = (np.stack([vals["landmarks"][n] for n in landmark_names])
                if landmark_names else np.zeros((0, fg.dimension)))
    pose_determined, lm_determined, ninfo = so.determined_masks(rp, u)
    res, ex = so.optimal_residuals(rp, u)
    obj_n = so.LiteralModel(fg, "SOCP").direct_cost(vals)
    n_first_chain = len(fg.pose_variables[0])
    assert pose_determined[:n_first_chain].all(), "the pinned robot's poses must be determined"
    print(f"{name}: newton iters {info['iters']} |g| {info['grad_inf']:.2e} obj {obj_n:.10f} | null space {ninfo['null_dim']} of "
          f"{ninfo['basis']} gauge directions | determined poses {int(pose_determined.sum())} of {len(pose_determined)}, "
          f"landmarks {lm_determined.tolist()} | active cones {int((ex > 1e-9).sum())} of {len(ex)}")
    # sanity line only: the ADMM twin on the same graph
    try:
        mdl = assemble(fg, "SOCP")
        sol = ConicSolver(mdl.qp, dict(eps_abs=1e-9, eps_rel=1e-9, max_iters=30000), lib_path=TWIN).solve()[0]
        Pa = mdl.pose_blocks(mdl.expand(sol.x))
        scale = max(1.0, float(np.max(np.abs(P_newton))))
        err = np.max(np.abs(Pa - P_newton).reshape(len(P_newton), -1), axis=1) / scale
        print(f"   (twin: status {sol.info['status']} pobj {sol.info['pobj']:.10f}; max pose difference on determined poses "
              f"{err[pose_determined].max():.2e}, on the others {err[~pose_determined].max() if (~pose_determined).any() else 0.0:.2e})")
    except Exception as exc:  # noqa: BLE001
        print("   (twin sanity run failed:", exc, ")")
    np.savez_compressed(
        os.path.join(HERE, f"{name}_golden.npz"),
        objective=np.float64(obj_n), poses=P_newton, landmarks=L_newton, landmark_determined=lm_determined,
        pose_determined=pose_determined, quad_residuals=res, quad_weights=rp.w, range_excess=ex,
        pose_names=np.array(pose_names, dtype="U"), landmark_names=np.array(landmark_names, dtype="U"),
        masks_from=np.array("oracle: null space of the generalised Hessian restricted to the odometry gauge basis"),
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
    golden_for("graph3d", graph_3d(n=40))
    golden_for("prior2d", graph_by_name("prior2d", {}))  # 2-D landmark priors (gurobi_utils.py:433-446)
