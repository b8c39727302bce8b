"""Back-to-back launch times of the K product and of empty launches on the headline problem (single-problem handle):
    python profiles/scripts/r04_probe.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from score_amd.manhattan import make_manhattan
from score_amd.native import assemble_native
from score_amd.solver import ConicSolver
m = assemble_native(make_manhattan(n_robots=20, n_poses=1000, n_beacons=4, seed=3000), "SOCP")
s = ConicSolver([m.qp], dict(polish=0, adaptive_cg=0))
s.steps(50)
print(" ".join(f"{k} {1e3 * s.debug_time(k, 300):.2f}" for k in ("nop1", "nop", "nop_load", "kp", "kpb", "rhs", "cone", "prec_init", "prec_step")), "us back to back", flush=True)
s.close()
