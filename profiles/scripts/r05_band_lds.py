"""A/B of the band tiles' operand window: global 16-byte loads per lane (SCORE_BAND_LDS=0) against the window staged in LDS
once per tile (SCORE_BAND_LDS=1).  In-loop dispatch times of the six kernels, single headline problem and a lock-step batch
of 16; the Newton polish's H product through the default solve.  python profiles/scripts/r05_band_lds.py [batch]"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import numpy as np
from score_amd.manhattan import make_manhattan
from score_amd.native import assemble_native
from score_amd.solver import ConicSolver

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
qps = [assemble_native(make_manhattan(n_robots=20, n_poses=1000, n_beacons=4, seed=3000 + j), "SOCP").qp for j in range(batch)]
ref = None
for mode in ("0", "1", "0", "1"):
    os.environ["SCORE_BAND_LDS"] = mode
    for name, sel in (("single", qps[:1]), (f"batch{batch}", qps)):
        s = ConicSolver(sel, dict(polish=0))
        dev, disp = s.time_iteration(warmup=50, iters=200, dispatch=True)
        s.close()
        print(f"LDS={mode} {name:8s} dispatch us: " + " ".join(f"{k} {v:6.2f}" for k, v in disp.items()) + f" | sum {sum(disp.values()):7.2f}", flush=True)
    s = ConicSolver(qps[:1], {})
    r = s.solve()[0]
    t = min(s.solve()[0].info["solve_ms"] for _ in range(5))
    s.close()
    if ref is None:
        ref = r.x
    print(f"LDS={mode} default solve {t:.3f} ms, newton {r.info['newton_iters']} / pcg {r.info['newton_cg_iters']}, x identical to first run: {np.array_equal(ref, r.x)}", flush=True)
