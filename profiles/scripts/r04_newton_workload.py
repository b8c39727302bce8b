"""Profiler workload for the Newton-polish launches: default solves of the headline problem (mode "headline") or of one
lock-step handle of 16 BASELINE configs[4] trials (mode "mc16").    python profiles/scripts/r04_newton_workload.py headline|mc16"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from score_amd.solver import ConicSolver
args = bench.parse_args([])
mode = sys.argv[1] if len(sys.argv) > 1 else "headline"
models = bench.make_headline(args, 0, 1) if mode == "headline" else bench.mc_models(args, range(16))
s = ConicSolver([m.qp for m in models], {})
for _ in range(3):
    outs = s.solve()
print(mode, "solved", sum(o.solved for o in outs), "newton", max(o.info["newton_iters"] for o in outs), "pcg", max(o.info["newton_cg_iters"] for o in outs))
s.close()
