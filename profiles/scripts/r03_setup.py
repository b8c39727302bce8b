"""Host-side phases of score_create on the headline graph: MINIMUM over repeated creations (the boxes' host cores are
shared; single timings scatter by 2-5x).  python profiles/scripts/r03_setup.py [reps [robots [batch]]]
(batch > 1: that many different graphs in one lock-step handle, e.g. `15 4 16` = a group of BASELINE configs[4] trials)"""
import os, re, sys, tempfile, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from score_amd.manhattan import make_manhattan
from score_amd.native import assemble_native, graph_arrays
from score_amd.solver import ConicSolver

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
robots = int(sys.argv[2]) if len(sys.argv) > 2 else 20
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
fgs = [make_manhattan(n_robots=robots, n_poses=1000, n_beacons=4, seed=3000 + t) for t in range(batch)]
arrs = [graph_arrays(fg) for fg in fgs]
ms = [assemble_native(fg, "SOCP", arrays=arr) for fg, arr in zip(fgs, arrs)]
ConicSolver([m.qp for m in ms], {}).close()
best, order, creates, assembles = {}, [], [], []
for i in range(reps):
    t = time.perf_counter(); ms = [assemble_native(fg, "SOCP", arrays=arr) for fg, arr in zip(fgs, arrs)]; assembles.append(time.perf_counter() - t)
    with tempfile.TemporaryFile(mode="w+") as tf:
        sys.stderr.flush()
        saved = os.dup(2); os.dup2(tf.fileno(), 2)
        try:
            t = time.perf_counter(); s = ConicSolver([m.qp for m in ms], dict(verbose=1)); creates.append(time.perf_counter() - t)
        finally:
            sys.stderr.flush(); os.dup2(saved, 2); os.close(saved)
        s.close()
        tf.seek(0)
        for line in tf.read().splitlines():
            mt = re.match(r"\[score setup\]\s+(.*?)\s+([0-9.]+) ms", line)
            if mt:
                k = mt.group(1).strip()
                if k not in best: order.append(k)
                best[k] = min(best.get(k, 1e9), float(mt.group(2)))
print(f"{batch} x ({robots} robots x 1000 poses): score_assemble: min {1e3*min(assembles):.2f} ms; score_create (from Python): min {1e3*min(creates):.2f} ms, median {1e3*sorted(creates)[len(creates)//2]:.2f} ms over {reps}")
for k in order:
    if not k.startswith("destroy"): print(f"  {k:40s} {best[k]:7.2f} ms")
