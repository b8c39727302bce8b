"""BASELINE configs[4] end to end from flat arrays: where a lock-step group's wall time goes (one thread,
then the 4-thread sweep), and the Python profile of one group."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cProfile, pstats
import numpy as np
from score_amd.manhattan import make_manhattan
from score_amd import solve_score as ss
from score_amd.native import ArrayGraph, graph_arrays, assemble_native
from score_amd.solver import ConicSolver

trials = [make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=5000 + t) for t in range(64)]
flat = [ArrayGraph(graph_arrays(fg)) for fg in trials]
st = dict(eps_abs=1e-7, eps_rel=1e-7)
ss.solve_score_batch(flat[:16], "SOCP", solver_settings=st)
def timed(fn):
    c, t = time.process_time(), time.perf_counter()
    out = fn()
    return out, 1e3 * (time.perf_counter() - t), 1e3 * (time.process_time() - c)
for rep in range(3):
    g = flat[:16]
    models, t_asm, c_asm = timed(lambda: [ss._model_for(d, "SOCP", "via_socp") for d in g])
    settings = dict(ss.DEFAULT_SOLVER_SETTINGS); settings.update(st)
    solver, t_create, c_create = timed(lambda: ConicSolver([m.qp for m in models], settings))
    sols, t_solve, c_solve = timed(solver.solve)
    out, t_extract, c_extract = timed(lambda: [ss.extract_solver_results(m, s.x, d, total_time=0.0, solved=s.solved, requested_relaxation="SOCP",
                                                                         info=s.info, lib=solver.lib, device=0) for d, m, s in zip(g, models, sols)])
    _, t_close, c_close = timed(solver.close)
    print(f"group of 16, one thread, wall ms (CPU ms of the whole process): assemble {t_asm:.1f} ({c_asm:.1f})  create {t_create:.1f} ({c_create:.1f})  "
          f"solve {t_solve:.1f} ({c_solve:.1f})  extract {t_extract:.1f} ({c_extract:.1f})  close {t_close:.1f} ({c_close:.1f})", flush=True)
for workers, gsz in ((1, None), (4, None), (8, None), (8, 16), (6, 11), (12, 8), (16, 4)):
    ts = []
    c0 = time.process_time()
    for _ in range(6):
        t = time.perf_counter(); r = ss.solve_score_batch(flat, "SOCP", solver_settings=st, workers=workers, group_size=gsz); ts.append(time.perf_counter() - t)
    cpu = time.process_time() - c0
    print(f"64 trials, workers {workers}, group size {gsz}: min {1e3*min(ts):.1f} ms = {64/min(ts):.0f} problems/s, median {1e3*sorted(ts)[3]:.1f} ms = {64/sorted(ts)[3]:.0f} problems/s, "
          f"solved {sum(x.solved for x in r)}; CPU time / wall time = {cpu/sum(ts):.1f} ({1e3*cpu/6/64:.1f} core-ms per trial)", flush=True)
pr = cProfile.Profile(); pr.enable()
ss.solve_score_batch(flat[:16], "SOCP", solver_settings=st)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
