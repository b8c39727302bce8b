"""Per-rank loads of the strong-scaling runs (64 trials over N GPUs -> 32 / 16 / 8 trials per rank): which grouping is fastest?"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from concurrent.futures import ThreadPoolExecutor
from score_amd.solver import ConicSolver
args0 = bench.parse_args([])
models = bench.mc_models(args0, range(32))
for n, sizes in ((32, [8] * 4), (32, [16, 16]), (32, [11, 11, 10]), (16, [4] * 4), (16, [8, 8]), (16, [16]), (16, [6, 5, 5]), (8, [4, 4]), (8, [8]), (8, [3, 3, 2])):
    solvers, o = [], 0
    for sz in sizes:
        solvers.append(ConicSolver([m.qp for m in models[o:o + sz]], dict(eps_abs=1e-7, eps_rel=1e-7))); o += sz
    with ThreadPoolExecutor(max_workers=len(sizes)) as pool:
        list(pool.map(lambda s: s.solve(), solvers))
        t0 = time.perf_counter()
        for _ in range(5):
            list(pool.map(lambda s: s.solve(), solvers))
        dt = (time.perf_counter() - t0) / 5
    print(f"{n} trials as {sizes}: {1e3*dt:.2f} ms per sweep, {n/dt:.0f} problems/s", flush=True)
    for s in solvers: s.close()
