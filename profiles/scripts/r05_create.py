"""score_create from a host-assembled program against score_create_from_graphs (model construction on the device), with the
setup's phase marks: MINIMUM over repeated creations.  python profiles/scripts/r05_create.py [reps [robots [batch]]]"""
import os, re, sys, tempfile, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from score_amd.manhattan import make_manhattan
from score_amd.native import assemble_native_batch, graph_arrays
from score_amd.solver import ConicSolver

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
robots = int(sys.argv[2]) if len(sys.argv) > 2 else 20
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
fgs = [make_manhattan(n_robots=robots, n_poses=1000, n_beacons=4, seed=3000 + t) for t in range(batch)]
arrs = [graph_arrays(fg) for fg in fgs]


def timed(make):
    best, order, times = {}, [], []
    make().close()
    for _ in range(reps):
        with tempfile.TemporaryFile(mode="w+") as tf:
            sys.stderr.flush()
            saved = os.dup(2); os.dup2(tf.fileno(), 2)
            try:
                t = time.perf_counter(); s = make(); times.append(time.perf_counter() - t)
            finally:
                sys.stderr.flush(); os.dup2(saved, 2); os.close(saved)
            s.close()
            tf.seek(0)
            for line in tf.read().splitlines():
                mt = re.match(r"\[score setup\]\s+(.*?)\s+([0-9.]+) ms", line)
                if mt and not mt.group(1).startswith("destroy"):
                    k = mt.group(1).strip()
                    if k not in best: order.append(k)
                    best[k] = min(best.get(k, 1e9), float(mt.group(2)))
    return times, best, order


def host_path():
    ms = assemble_native_batch(arrs, "SOCP")
    return ConicSolver([m.qp for m in ms], dict(verbose=1))


for name, make in (("score_assemble_batch + score_create", host_path), ("score_create_from_graphs", lambda: ConicSolver.from_graphs(arrs, 0, dict(verbose=1)))):
    times, best, order = timed(make)
    print(f"{batch} x ({robots} robots x 1000 poses): {name}: min {1e3*min(times):.2f} ms, median {1e3*sorted(times)[len(times)//2]:.2f} ms over {reps}")
    for k in order:
        print(f"  {k:52s} {best[k]:7.2f} ms")
