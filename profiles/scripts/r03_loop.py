"""In-loop kernel times of the ADMM iteration (dispatch durations), single headline problem and a lock-step batch,
plus iterations/s of the ADMM loop alone and the product default solve.
    python profiles/scripts/r03_loop.py [batch=16] [lib=<path to an alternative libscore_hip.so>]"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from score_amd.manhattan import make_manhattan
from score_amd.native import assemble_native
from score_amd.solver import ConicSolver

B, lib = 16, None
for a in sys.argv[1:]:
    if a.startswith("batch="): B = int(a[6:])
    if a.startswith("lib="): lib = a[4:]
models = [assemble_native(make_manhattan(n_robots=20, n_poses=1000, n_beacons=4, seed=3000 + j), "SOCP") for j in range(max(1, B))]
names = ConicSolver.ITERATION_KERNELS
for nb in (1, B):
    if nb < 1: continue
    s = ConicSolver([m.qp for m in models[:nb]], dict(polish=0, adaptive_cg=0), lib_path=lib)
    dev, disp = s.time_iteration(warmup=20, iters=100 if nb == 1 else 40, dispatch=True)
    _, kb = s.time_kkt_apply(10)
    tot = sum(disp.values())
    print(f"batch {nb}: " + " ".join(f"{k} {disp[k]:.2f}" for k in names) + f" | sum {tot:.1f} us, per problem {tot/nb:.2f} us, "
          f"kkt {kb/1e6:.2f} MB -> {kb/disp['kp']/1e6:.3f} TB/s ({kb/disp['kp']/8e6:.3f} of 8 TB/s), rep {int(s.debug_get('rep')[0])}", flush=True)
    print(f"   device clock: " + " ".join(f"{k} {dev[k]:.2f}" for k in names), flush=True)
    if nb == 1:
        s.solve()
        t0 = time.perf_counter(); its = 0
        for _ in range(5):
            its += s.solve()[0].info["iters"]
        dt = time.perf_counter() - t0
        print(f"   ADMM loop alone: {its/dt:.0f} it/s ({its//5} iterations per solve, {1e3*dt/5:.2f} ms)", flush=True)
    s.close()
s = ConicSolver(models[0].qp, {}, lib_path=lib)
s.solve()
t0 = time.perf_counter()
for _ in range(10):
    o = s.solve()[0]
print(f"product default: {1e2*(time.perf_counter()-t0):.2f} ms, admm {o.info['iters']} newton {o.info['newton_iters']} pcg {o.info['newton_cg_iters']} solved {o.solved}")
s.close()
