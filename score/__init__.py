"""Import-path shim: ``from score.solve_score import solve_score`` resolves to the
MI355X implementation in ``score_amd`` (drop-in for MarineRoboticsGroup/score)."""
