"""Two builds of the library against each other, bit by bit (x, y of fixed-length ADMM runs and of default solves; headline problem,
a 3000-pose chain, a lock-step batch of four).  The builds compared: commit 02a32da (before the first-trip work on the kernels, the
polling mode of the split rows and the queue-ahead of the Newton control fetch were in) against HEAD; edit `libs` to compare others
(build one with: git archive <commit> score_amd/csrc include | tar -x -C /tmp/x && hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC
-I/tmp/x/include -o lib.so /tmp/x/score_amd/csrc/score_hip.hip).  python r05_cmp_libraries.py"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from score_amd.native import assemble_native
from score_amd.manhattan import make_manhattan
from score_amd.solver import ConicSolver
libs = ["scratch/bisect/02a32da/libscore_hip.so", "score_amd/csrc/libscore_hip.so"]
cases = {
  "headline": [assemble_native(make_manhattan(n_robots=20, n_poses=1000, n_beacons=4, seed=0), "SOCP").qp],
  "chain3000": [assemble_native(make_manhattan(n_robots=1, n_poses=3000, n_beacons=3, seed=8), "SOCP").qp],
  "batch4": [assemble_native(make_manhattan(n_robots=3, n_poses=300 + 50 * k, n_beacons=3, seed=40 + k), "SOCP").qp for k in range(4)],
}
for name, qps in cases.items():
    for st in (dict(polish=0, max_iters=200, eps_abs=1e-30, eps_rel=1e-30, check_interval=50, adaptive_rho=0, adaptive_cg=0), {}):
        out = []
        for lib in libs:
            sv = ConicSolver(qps, st, lib_path=lib)
            r = sv.solve()
            out.append(r); sv.close()
        eq = all(np.array_equal(a.x, b.x) and np.array_equal(a.y, b.y) for a, b in zip(*out))
        md = max(float(np.abs(a.x - b.x).max()) for a, b in zip(*out))
        print(name, "default" if not st else "admm200", "bit-equal" if eq else "DIFFERENT", "max |dx| %.3e" % md, [x.info["iters"] for x in out[0]], [x.info["iters"] for x in out[1]], flush=True)
