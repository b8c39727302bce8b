"""Stress run from seeds: 128 worlds of BASELINE configs[4]'s shape and 500 worlds of random shapes (1-6 robots, 20-1600 poses --
chains beyond 1023 poses take the segmented chain kernel --, 0-6 beacons, measurement probability 0.02-0.5, batches of 1-8) drawn on the
device, solved as SOCP and as direct QCQP (objectives must agree), world 0 of every batch certified through the array path.
python profiles/scripts/r05_stress.py"""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
from score_amd.generate import GeneratedBatch
from score_amd.solve_score import solve_score_batch, solve_score
from score_amd.native import assemble_native
from score_amd.solver import ConicSolver
from oracle import score_oracle as so
rng = np.random.default_rng(2025)
bad = []
t0 = time.time()
# 1. BASELINE configs[4]'s shape, 512 worlds
for k in range(2):
    B = GeneratedBatch(64, seed=100000 + 64 * k, n_robots=4, n_poses=1000, n_beacons=4)
    rs = solve_score_batch(B.graphs(), "SOCP")
    ns = sum(r.solved for r in rs)
    worst = max(max(v["translation_max"] for v in B.trajectory_errors(i, rs[i]).values()) for i in range(64))
    print(f"batch {k}: solved {ns}/64, max newton {max(r.info['newton_iters'] for r in rs)}, worst translation error {worst:.2f} m, {time.time()-t0:.1f} s", flush=True)
    if ns != 64: bad.append(("cfg4", k))
# 2. random shapes
for trial in range(500):
    R = int(rng.integers(1, 7)); T = int(rng.integers(20, 1600)); Nb = int(rng.integers(0 if R > 1 else 1, 7))
    p = float(rng.uniform(0.02, 0.5)); side = int(rng.integers(3, 40))
    cnt = int(rng.integers(1, 9))
    B = GeneratedBatch(cnt, seed=int(rng.integers(0, 2**40)), n_robots=R, n_poses=T, n_beacons=Nb, side=side, p_range=p)
    gs = B.graphs()
    for relax, mode in (("SOCP", "via_socp"), ("QCQP", "direct")):
        try:
            rs = solve_score_batch(gs, relax, qcqp_mode=mode)
        except Exception as exc:
            bad.append((trial, R, T, Nb, relax, repr(exc)[:200])); print("EXC", bad[-1], flush=True); continue
        if not all(r.solved for r in rs):
            bad.append((trial, R, T, Nb, p, side, relax, [r.info["status"] for r in rs])); print("UNSOLVED", bad[-1], flush=True)
        if relax == "SOCP":
            obj = [r.info["pobj"] for r in rs]
        else:
            for a, b in zip(obj, [r.info["pobj"] for r in rs]):
                if abs(a - b) > 1e-6 * max(1.0, abs(a)):
                    bad.append((trial, "objective", a, b)); print("OBJ", bad[-1], flush=True)
    # KKT certificate of world 0 through the array path
    a0 = B.arrays(0)
    qp = assemble_native(gs[0], "SOCP", arrays=a0).qp
    sv = ConicSolver([qp], {}); out = sv.solve()[0]; sv.close()
    cert = so.kkt_certificate(qp.P, qp.q, qp.A, qp.b, 0, qp.soc_dims, out.x, out.y, out.s)
    if not (out.solved and cert["primal_res_inf"] < 1e-5 and cert["dual_res_inf"] < 1e-4):
        bad.append((trial, "cert", R, T, Nb, cert)); print("CERT", bad[-1], flush=True)
    if trial % 100 == 99: print(f"random shapes: {trial+1} done, {len(bad)} bad, {time.time()-t0:.1f} s", flush=True)
print("BAD:", bad)
