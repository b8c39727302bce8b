"""``ScoreSolverParams`` as the reference's example expects it
(examples/solve_goats_example_score.py:21,28-34; the module is missing from the
reference tree).  The convex solve needs no initialisation, so the fields are
carried for compatibility only."""
from dataclasses import dataclass
from typing import Optional


@dataclass
class ScoreSolverParams:
    solver: str = "hip"
    verbose: bool = False
    save_results: bool = False
    init_technique: str = "none"
    custom_init_file: Optional[str] = None
