"""Where a from-arrays sweep of BASELINE configs[4] (64 fresh graphs through solve_score_batch, model construction on the
device) spends its wall time: every score_create_from_graphs, solve + read-back and close with its thread and interval, host CPU
per problem.  python profiles/scripts/r05_e2e_timeline.py [workers [group_size]]"""
import os, resource, sys, threading, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import score_amd.solve_score as S
from score_amd.manhattan import make_manhattan
from score_amd.native import ArrayGraph, graph_arrays
from score_amd.solver import ConicSolver

workers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
group = int(sys.argv[2]) if len(sys.argv) > 2 else None
log = []
def wrap(obj, name, label, cm=False):
    f = getattr(obj, name)
    f = f.__func__ if cm else f
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            log.append((label, threading.get_ident(), t, time.perf_counter()))
    setattr(obj, name, classmethod(g) if cm else g)
wrap(S, "_models_for", "models")
wrap(ConicSolver, "from_graphs", "create", cm=True)
wrap(ConicSolver, "solve_estimates", "solve+read")
wrap(ConicSolver, "close", "close")
trials = [make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=5000 + t) for t in range(64)]
flat = [ArrayGraph(graph_arrays(fg)) for fg in trials]
st = dict(device=0)
for _ in range(2):
    S.solve_score_batch(flat, "SOCP", solver_settings=st, workers=workers, group_size=group)
best = None
walls = []
for rep in range(8):
    log.clear()
    r0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    rs = S.solve_score_batch(flat, "SOCP", solver_settings=st, workers=workers, group_size=group)
    wall = time.perf_counter() - t0
    r1 = resource.getrusage(resource.RUSAGE_SELF)
    cpu = 1e3 * ((r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime))
    walls.append(wall)
    print(f"  sweep {rep}: {1e3*wall:.1f} ms; host cpu {cpu:.0f} ms = {cpu/64:.2f} ms per problem (user {1e3*(r1.ru_utime-r0.ru_utime):.0f} sys {1e3*(r1.ru_stime-r0.ru_stime):.0f}); "
          f"minor faults {r1.ru_minflt-r0.ru_minflt}; ctx switches {r1.ru_nvcsw-r0.ru_nvcsw}+{r1.ru_nivcsw-r0.ru_nivcsw}")
    if best is None or wall < best[0]:
        best = (wall, t0, list(log))
wall, t0, lg = best
print(f"workers {workers} group {group}: best sweep {1e3*wall:.1f} ms = {64/wall:.0f} graphs/s, median {1e3*sorted(walls)[len(walls)//2]:.1f} ms = {64/sorted(walls)[len(walls)//2]:.0f} graphs/s, solved {sum(r.solved for r in rs)}")
cats = {}
for lab, tid, a, b in lg:
    c = cats.setdefault(lab, [0, 0.0, 1e9, 0.0])
    c[0] += 1; c[1] += b - a; c[2] = min(c[2], a - t0); c[3] = max(c[3], b - t0)
for lab, (n, tot, first, last) in cats.items():
    print(f"  {lab:10s} {n:4d} calls  sum {1e3*tot:8.1f} ms  avg {1e3*tot/n:7.2f} ms  first start {1e3*first:6.1f}  last end {1e3*last:6.1f}")
tids = sorted({t for _, t, _, _ in lg})
for tid in tids:
    ev = sorted((a - t0, b - t0, lab) for lab, t, a, b in lg if t == tid)
    print(f"  thread {tids.index(tid)}: " + " ".join(f"{lab[:2]}[{1e3*a:.1f}-{1e3*b:.1f}]" for a, b, lab in ev))
