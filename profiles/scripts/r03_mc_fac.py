"""BASELINE configs[4] on one GPU (64 trials, 4 x 16 lock-step handles) under the factor precision modes."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from concurrent.futures import ThreadPoolExecutor
import bench
from score_amd.solver import ConicSolver

args = bench.parse_args([])
models = bench.mc_models(args, range(64))
for mode in (1, 2, 1, 2):
    solvers = [ConicSolver([m.qp for m in models[o:o + 16]], dict(fac_fp32=mode)) for o in range(0, 64, 16)]
    with ThreadPoolExecutor(max_workers=4) as pool:
        list(pool.map(lambda s: s.solve(), solvers))
        t0 = time.perf_counter()
        for _ in range(4):
            outs = list(pool.map(lambda s: s.solve(), solvers))
        dt = time.perf_counter() - t0
    pcg = sum(o.info["newton_cg_iters"] for g in outs for o in g) / 64
    nit = sum(o.info["newton_iters"] for g in outs for o in g) / 64
    print(f"fac_fp32 {mode}: {64*4/dt:.0f} problems/s, newton {nit:.1f}, pcg per problem {pcg:.1f}, solved {sum(o.solved for g in outs for o in g)}", flush=True)
    for s in solvers: s.close()
