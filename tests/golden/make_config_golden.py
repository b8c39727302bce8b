"""Oracle optimum of a BASELINE config at its FULL size (run in the build container only).

    python tests/golden/make_config_golden.py 3        # 20 robots x 1000 poses, the headline config

`manhattan.make_config(index)` regenerates the graph from its seed on any machine, so only the
oracle's OUTPUT travels: objective, every pose block, landmark positions, which poses / landmarks
the optimum determines, and the residuals every optimum shares -- the same fields as the
`*_golden.npz` files of make_fixture_golden.py, from the oracle alone (semismooth Newton with SuperLU,
oracle/score_oracle.py; the product's assembler, setup and solvers are not involved).

The pose-by-pose comparison north_star states (1e-4 relative on what
/root/reference/score/solve_score.py:54-86 returns) then runs on the GPU box against this file:
tests/test_gpu_parity.py::test_full_size_configs_are_certified.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import score_oracle as so  # noqa: E402
from score_amd.manhattan import make_config  # noqa: E402


def main(index: int) -> None:
    t0 = time.time()
    fg = make_config(index)
    rp, u, info = so.newton_solve(fg, tol=1e-13, max_iter=300, verbose=True)
    print(f"config {index}: newton iters {info['iters']} |g| {info['grad_inf']:.3e} objective {info['objective']:.12f} "
          f"({time.time() - t0:.0f} s)", flush=True)
    vals = so.reduced_to_values(rp, u, "SOCP")
    pose_names = [p.name for chain in fg.pose_variables for p in chain]
    landmark_names = [l.name for l in fg.landmark_variables]
    poses = np.stack([vals["poses"][n] for n in pose_names])
    landmarks = (np.stack([vals["landmarks"][n] for n in landmark_names])
                 if landmark_names else np.zeros((0, fg.dimension)))
    pose_determined, lm_determined, ninfo = so.determined_masks(rp, u)
    res, ex = so.optimal_residuals(rp, u)
    objective = so.LiteralModel(fg, "SOCP").direct_cost(vals)
    print(f"   literal objective {objective:.12f}; null space {ninfo['null_dim']} of {ninfo['basis']} gauge directions; "
          f"determined poses {int(pose_determined.sum())} of {len(pose_determined)}, landmarks {lm_determined.tolist()}; "
          f"active cones {int((ex > 1e-9).sum())} of {len(ex)} ({time.time() - t0:.0f} s)", flush=True)
    np.savez_compressed(
        os.path.join(HERE, f"config{index}_golden.npz"),
        objective=np.float64(objective), poses=poses.astype(np.float64), landmarks=landmarks,
        landmark_determined=lm_determined, pose_determined=pose_determined,
        quad_residuals=res, quad_weights=rp.w, range_excess=ex,
        pose_names=np.array(pose_names, dtype="U"), landmark_names=np.array(landmark_names, dtype="U"),
        grad_inf=np.float64(info["grad_inf"]), newton_iters=np.int64(info["iters"]),
        masks_from=np.array("oracle: null space of the generalised Hessian restricted to the odometry gauge basis"),
    )


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
