"""Default solver against the number of ADMM warm-up iterations: headline problem and 12 config-5 trials."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from score_amd.manhattan import make_manhattan
from score_amd.native import assemble_native, graph_arrays
from score_amd.solver import ConicSolver
def qp_of(r, seed):
    fg = make_manhattan(n_robots=r, n_poses=1000, n_beacons=4, seed=seed)
    return assemble_native(fg, "SOCP", arrays=graph_arrays(fg))
head = qp_of(20, 3000)
small = [qp_of(4, 5000 + t) for t in range(12)]
for wu in (4, 6, 8, 10, 12, 15, 20):
    s = ConicSolver([head.qp], dict(polish_warmup=wu)); s.solve()
    best = min((s.solve()[0] for _ in range(4)), key=lambda o: o.info["solve_ms"]); s.close()
    tot = 0.0; nit = 0; pcg = 0; ok = 0
    for m in small:
        s = ConicSolver([m.qp], dict(polish_warmup=wu)); s.solve()
        o = min((s.solve()[0] for _ in range(3)), key=lambda o: o.info["solve_ms"]); s.close()
        tot += o.info["solve_ms"]; nit += o.info["newton_iters"]; pcg += o.info["newton_cg_iters"]; ok += o.solved
    print(f"warm-up {wu:2d}: headline {best.info['solve_ms']:.2f} ms ({best.info['newton_iters']} Newton, {best.info['newton_cg_iters']} PCG, solved {best.solved}); "
          f"12 config-5 trials {tot:.1f} ms ({nit} Newton, {pcg} PCG, {ok} solved)", flush=True)
