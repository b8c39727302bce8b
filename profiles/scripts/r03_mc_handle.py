"""One lock-step handle of 16 (and 4, 8, 32) configs[4] trials alone on the GPU: solve time, Newton / PCG counts;
then the verbose Newton timeline of the 16-trial handle."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from score_amd.solver import ConicSolver
args = bench.parse_args([])
models = bench.mc_models(args, range(32))
for nb in (1, 4, 8, 16, 32):
    s = ConicSolver([m.qp for m in models[:nb]], {})
    s.solve()
    t0 = time.perf_counter()
    for _ in range(5): outs = s.solve()
    dt = (time.perf_counter() - t0) / 5
    print(f"{nb:2d} trials in one handle: {1e3*dt:.2f} ms per solve ({nb/dt:.0f} problems/s alone), newton max {max(o.info['newton_iters'] for o in outs)}, "
          f"pcg max {max(o.info['newton_cg_iters'] for o in outs)}, admm {outs[0].info['iters']}", flush=True)
    s.close()
s = ConicSolver([m.qp for m in models[:16]], dict(verbose=1))
s.solve(); s.close()
