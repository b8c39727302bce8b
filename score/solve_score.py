"""Same import path as the reference's score/solve_score.py."""
from score_amd.solve_score import (  # noqa: F401
    _check_factor_graph,
    solve_problem_with_intermediate_iterates,
    solve_score,
    solve_score_batch,
)
