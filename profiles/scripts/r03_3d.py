"""3-D problems (4 robots x 1000 poses, 4 beacons): in-loop kernel times with the LDS-resident chain kernel for 4 x 4
blocks (fac_fp32 = 1, default) and with the streaming one (fac_fp32 = 0), iterations/s, product default solve."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from score_amd.manhattan import make_manhattan_3d
from score_amd.native import assemble_native
from score_amd.solver import ConicSolver

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
m = assemble_native(make_manhattan_3d(n_robots=R, n_poses=1000, n_beacons=4, seed=7000), "SOCP")
print(f"3-D {R} x 1000: n {m.qp.n} m {m.qp.m} nnz(P) {m.qp.P.nnz}")
names = ConicSolver.ITERATION_KERNELS
for fp32 in (1, 2, 0):
    s = ConicSolver([m.qp], dict(polish=0, adaptive_cg=0, fac_fp32=fp32))
    dev, disp = s.time_iteration(warmup=20, iters=100, dispatch=True)
    _, kb = s.time_kkt_apply(10)
    print(f"fac_fp32 {fp32}: " + " ".join(f"{k} {disp[k]:.2f}" for k in names) + f" | sum {sum(disp.values()):.1f} us, kkt {kb/1e6:.2f} MB, rep {int(s.debug_get('rep')[0])}", flush=True)
    s.solve(); t0 = time.perf_counter(); o = s.solve()[0]; dt = time.perf_counter() - t0
    print(f"   ADMM alone: {o.info['iters']/dt:.0f} it/s ({o.info['iters']} iterations, {1e3*dt:.1f} ms, solved {o.solved})", flush=True)
    s.close()
    s = ConicSolver([m.qp], dict(fac_fp32=fp32)); s.solve(); t0 = time.perf_counter(); o = s.solve()[0]; dt = time.perf_counter() - t0
    print(f"   default: {1e3*dt:.2f} ms admm {o.info['iters']} newton {o.info['newton_iters']} pcg {o.info['newton_cg_iters']} solved {o.solved} pobj {o.info['pobj']:.9f}", flush=True)
    s.close()
