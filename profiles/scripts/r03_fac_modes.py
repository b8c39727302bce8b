"""Product default solve under fac_fp32 = 1 (Newton factors double) and 2 (float stream: register-resident chain kernel
for the Newton PCG too) on the BASELINE sizes, and BASELINE configs[4] on one GPU."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from concurrent.futures import ThreadPoolExecutor
import bench
from score_amd.manhattan import make_manhattan
from score_amd.native import assemble_native
from score_amd.solver import ConicSolver

for (r, n, b, seed) in ((20, 1000, 4, 3000), (4, 1000, 4, 4000), (1, 500, 2, 1000), (3, 60, 3, 5)):
    qp = assemble_native(make_manhattan(n_robots=r, n_poses=n, n_beacons=b, seed=seed), "SOCP").qp
    for mode in (1, 2):
        s = ConicSolver(qp, dict(fac_fp32=mode)); s.solve()
        t0 = time.perf_counter()
        for _ in range(10): o = s.solve()[0]
        dt = (time.perf_counter() - t0) / 10
        print(f"{r}x{n} fac_fp32 {mode}: {dt*1e3:.2f} ms newton {o.info['newton_iters']} pcg {o.info['newton_cg_iters']} solved {o.solved} pobj {o.info['pobj']:.9f}", flush=True)
        s.close()
args = bench.parse_args([])
models = bench.mc_models(args, range(64))
for mode in (1, 2, 1, 2):
    solvers = [ConicSolver([m.qp for m in models[o:o + 16]], dict(fac_fp32=mode)) for o in range(0, 64, 16)]
    with ThreadPoolExecutor(max_workers=4) as pool:
        list(pool.map(lambda s: s.solve(), solvers))
        t0 = time.perf_counter()
        for _ in range(4): outs = list(pool.map(lambda s: s.solve(), solvers))
        dt = time.perf_counter() - t0
    print(f"MC fac_fp32 {mode}: {64*4/dt:.0f} problems/s, pcg per problem {sum(o.info['newton_cg_iters'] for g in outs for o in g)/64:.1f}, solved {sum(o.solved for g in outs for o in g)}", flush=True)
    for s in solvers: s.close()
