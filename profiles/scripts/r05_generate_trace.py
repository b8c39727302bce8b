"""Kernel-trace target: batches of 64 synthetic Manhattan worlds (4 robots x 1000 poses, 4 beacons) drawn on the device
(score_generate_manhattan), then one sweep seeds -> estimates.  python r05_generate_trace.py [batches]"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from score_amd.generate import GeneratedBatch
from score_amd.solve_score import solve_score_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for k in range(n):
    t = time.perf_counter()
    B = GeneratedBatch(64, seed=9000 + 100 * k, n_robots=4, n_poses=1000, n_beacons=4)
    t1 = time.perf_counter()
    gs = B.graphs()
    print("generate 64 x (4 x 1000): %.2f ms library + %.2f ms views" % (1e3 * (t1 - t), 1e3 * (time.perf_counter() - t1)), flush=True)
solve_score_batch(gs, "SOCP")
for k in range(3):
    t = time.perf_counter()
    rs = solve_score_batch(GeneratedBatch(64, seed=20000 + 100 * k, n_robots=4, n_poses=1000, n_beacons=4).graphs(), "SOCP")
    dt = time.perf_counter() - t
    print("seeds -> estimates, 64 worlds: %.1f ms = %.0f/s, solved %d" % (1e3 * dt, 64 / dt, sum(r.solved for r in rs)), flush=True)
