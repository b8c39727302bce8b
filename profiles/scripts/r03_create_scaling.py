"""How concurrent score_create calls scale: k threads, each creating (and destroying) a 16-trial lock-step handle."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from concurrent.futures import ThreadPoolExecutor
from score_amd.manhattan import make_manhattan
from score_amd.native import assemble_native, graph_arrays
from score_amd.solver import ConicSolver
if os.environ.get("SWITCH"): sys.setswitchinterval(float(os.environ["SWITCH"]))
fgs = [make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=5000 + t) for t in range(64)]
arrs = [graph_arrays(fg) for fg in fgs]
ms = [assemble_native(fg, "SOCP", arrays=a) for fg, a in zip(fgs, arrs)]
groups = [[m.qp for m in ms[i:i + 16]] for i in range(0, 64, 16)]
def create(g):
    t = time.perf_counter(); s = ConicSolver(g, {}); dt = time.perf_counter() - t; s.close(); return dt
def assemble(i):
    t = time.perf_counter(); assemble_native(fgs[i], "SOCP", arrays=arrs[i]); return time.perf_counter() - t
for g in groups: create(g)
for k in (1, 2, 4):
    best = None
    for _ in range(5):
        c0 = time.process_time(); t = time.perf_counter()
        with ThreadPoolExecutor(max_workers=k) as pool: each = list(pool.map(create, groups[:k]))
        wall = time.perf_counter() - t; cpu = time.process_time() - c0
        if best is None or wall < best[0]: best = (wall, cpu, each)
    print(f"{k} concurrent creates of 16 trials: wall {1e3*best[0]:.1f} ms, CPU {1e3*best[1]:.1f} ms, each {[round(1e3*x,1) for x in best[2]]}", flush=True)
for k in (1, 4, 8, 16):
    best = None
    for _ in range(5):
        c0 = time.process_time(); t = time.perf_counter()
        with ThreadPoolExecutor(max_workers=k) as pool: each = list(pool.map(assemble, range(64)))
        wall = time.perf_counter() - t; cpu = time.process_time() - c0
        if best is None or wall < best[0]: best = (wall, cpu, each)
    print(f"64 model constructions on {k} threads: wall {1e3*best[0]:.1f} ms, CPU {1e3*best[1]:.1f} ms, mean call {1e3*sum(best[2])/64:.2f} ms", flush=True)
