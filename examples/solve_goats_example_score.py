"""Corrected twin of the reference's examples/solve_goats_example_score.py (which
is stale: it imports a module that does not exist and calls solve_score with
three positional arguments).  BASELINE.json config 0: the GOATS AUV data set.

    python examples/solve_goats_example_score.py [path/to/goats.pkl | fixture.npz]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from score.solve_score import solve_score  # noqa: E402
from score.utils.gurobi_utils import QCQP_RELAXATION  # noqa: E402
from score.utils.solver_utils import ScoreSolverParams  # noqa: E402
from score_amd.io import load_fg_npz, load_pyfg_pickle, save_to_tum  # noqa: E402
from score_amd.refine import refine_estimate  # noqa: E402

if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    default = os.path.join(os.path.dirname(here), "tests", "golden", "goats_fg.npz")
    path = sys.argv[1] if len(sys.argv) > 1 else default
    goats_pyfg = load_fg_npz(path) if path.endswith(".npz") else load_pyfg_pickle(path)
    solver_params = ScoreSolverParams(solver="hip", verbose=True)
    score_result = solve_score(goats_pyfg, solver_params, QCQP_RELAXATION)
    print(f"solved={score_result.solved} objective={score_result.solver_cost:.6f} "
          f"iterations={score_result.info['iters']} time={score_result.total_time:.3f}s")
    print("trajectory written to", save_to_tum(score_result, "/tmp/goats_score"))
    # the step the reference's README describes next (README.md:63-67): SCORE's estimate initialises a local
    # maximum-likelihood refinement -- here Gauss-Newton on SE(2), on the same GPU
    refined, info = refine_estimate(goats_pyfg, score_result)
    print(f"refined: cost {info['cost_initial']:.4f} -> {info['cost_final']:.4f} in {info['iterations']} iterations "
          f"({info['pcg_iters']} PCG iterations, {info['solve_ms']:.1f} ms)")
    print("refined trajectory written to", save_to_tum(refined, "/tmp/goats_refined"))
