"""BASELINE configs[4] as a Monte-Carlo study runs it: every graph solved ONCE -- handle creation (model construction
included) inside the timer.  64 four-robot graphs as flat arrays; groups of G graphs per lock-step handle on T host threads:
score_create_from_graphs (device) against score_assemble_batch + score_create (host assembler).
python profiles/scripts/r05_fresh.py [sweeps]"""
import os, resource, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import numpy as np
from score_amd.manhattan import make_manhattan
from score_amd.native import assemble_native_batch, graph_arrays
from score_amd.solver import ConicSolver

sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N = 64
arrs = [graph_arrays(make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=4000 + t)) for t in range(N)]
st = dict(eps_abs=1e-7, eps_rel=1e-7)


def one_device(idx):
    s = ConicSolver.from_graphs([arrs[i] for i in idx], 0, st)
    try:
        return s.solve()
    finally:
        s.close()


def one_host(idx):
    ms = assemble_native_batch([arrs[i] for i in idx], "SOCP")
    s = ConicSolver([m.qp for m in ms], st)
    try:
        return s.solve()
    finally:
        s.close()


def cpu_ms():
    ru = resource.getrusage(resource.RUSAGE_SELF)
    return 1e3 * (ru.ru_utime + ru.ru_stime)


for name, fn in (("from_graphs", one_device), ("host assemble", one_host)):
    for G, T in ((16, 4), (8, 8), (8, 4), (4, 8), (4, 16), (32, 2), (64, 1)):
        groups = [list(range(i, min(N, i + G))) for i in range(0, N, G)]
        with ThreadPoolExecutor(max_workers=T) as pool:
            list(pool.map(fn, groups))  # untimed
            best, cpu = 1e9, 0.0
            solved = 0
            for _ in range(sweeps):
                c0 = cpu_ms(); t0 = time.perf_counter()
                res = [r for rs in pool.map(fn, groups) for r in rs]
                dt = time.perf_counter() - t0
                if dt < best:
                    best, cpu = dt, cpu_ms() - c0
                solved = sum(r.solved for r in res)
        print(f"{name:14s} groups of {G:2d} on {T:2d} threads: best sweep {1e3*best:6.1f} ms = {N/best:6.0f} graphs/s, host CPU {cpu/N:5.2f} ms per graph, solved {solved}", flush=True)
