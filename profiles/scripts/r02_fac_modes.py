"""fac_fp32 = 0 / 1 (ADMM factors as floats) / 2 (Newton factors too): product default solve on the BASELINE sizes
and BASELINE configs[4] on one GPU."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from concurrent.futures import ThreadPoolExecutor
import bench
from score_amd.assemble import assemble
from score_amd.manhattan import make_manhattan
from score_amd.solver import ConicSolver
for (r, n, b, seed) in ((20, 1000, 4, 3000), (4, 1000, 4, 4000)):
    qp = assemble(make_manhattan(n_robots=r, n_poses=n, n_beacons=b, seed=seed), "SOCP").qp
    for mode in (0, 1, 2):
        s = ConicSolver(qp, dict(fac_fp32=mode)); s.solve()
        t0 = time.perf_counter()
        for _ in range(10): o = s.solve()[0]
        dt = (time.perf_counter() - t0) / 10
        i = o.info
        print(f"{r}x{n} fac_fp32={mode}: {dt*1e3:.2f} ms solved={o.solved} admm={i['iters']} newton={i['newton_iters']} pcg={i['newton_cg_iters']}", flush=True)
        s.close()
args = bench.parse_args([])
models = bench.mc_models(args, range(64))
for mode in (0, 1, 2, 0, 1, 2, 0, 1, 2):
    solvers = [ConicSolver([m.qp for m in models[o:o + 16]], dict(eps_abs=1e-7, eps_rel=1e-7, fac_fp32=mode)) for o in range(0, 64, 16)]
    with ThreadPoolExecutor(max_workers=4) as pool:
        list(pool.map(lambda s: s.solve(), solvers))
        t0 = time.perf_counter()
        for _ in range(20): outs = list(pool.map(lambda s: s.solve(), solvers))
        dt = (time.perf_counter() - t0) / 20
    pcg = sum(o[0].info["newton_cg_iters"] for o in outs)
    print(f"config 5, fac_fp32={mode}: {64/dt:.0f} problems/s, lock-step Newton PCG iterations {pcg}, solved {sum(x.solved for o in outs for x in o)}", flush=True)
    for s in solvers: s.close()
