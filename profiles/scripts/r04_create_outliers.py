"""solve_score() on the headline graph, call after call: score_create / solve / total per call, and the setup phases of the
calls whose create took more than 1.5 x the median.  python profiles/scripts/r04_create_outliers.py [calls]"""
import os, re, sys, tempfile, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from score_amd.manhattan import make_manhattan
from score_amd.solve_score import solve_score
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
fg = make_manhattan(n_robots=20, n_poses=1000, n_beacons=4, seed=3000)
st = dict(device=0, verbose=1)
solve_score(fg, "SOCP", solver_settings=dict(device=0))
rows = []
for i in range(n):
    with tempfile.TemporaryFile(mode="w+") as tf:
        sys.stderr.flush(); saved = os.dup(2); os.dup2(tf.fileno(), 2)
        try:
            t = time.perf_counter(); r = solve_score(fg, "SOCP", solver_settings=st); tot = time.perf_counter() - t
        finally:
            sys.stderr.flush(); os.dup2(saved, 2); os.close(saved)
        tf.seek(0); text = tf.read()
    rows.append((1e3 * tot, r.info["setup_ms"], r.info["solve_ms"], text))
med = sorted(x[1] for x in rows)[n // 2]
for i, (tot, cr, so, text) in enumerate(rows):
    print(f"call {i:2d}: total {tot:6.1f} ms  create {cr:6.2f}  solve {so:5.2f}")
    if cr > 1.5 * med:
        for line in text.splitlines():
            mt = re.match(r"\[score setup\]\s+(.*?)\s+([0-9.]+) ms\s+\((\d+) page faults, (\d+) context", line)
            if mt and float(mt.group(2)) > 0.8 and not mt.group(1).startswith("destroy"):
                print(f"      {mt.group(1):40s} {float(mt.group(2)):7.2f} ms  faults {mt.group(3)} ctx {mt.group(4)}")
