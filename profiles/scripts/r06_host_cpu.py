"""Where the host CPU of a fresh-graph Monte-Carlo sweep goes, thread by thread: 64 generated worlds per sweep in 4 lock-step
handles of 16 on 4 host threads (the bench's fresh_graphs leg), /proc/self/task/*/stat sampled before and after N sweeps.
python profiles/scripts/r06_host_cpu.py [sweeps]"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from concurrent.futures import ThreadPoolExecutor
import numpy as np
from score_amd.manhattan import make_manhattan
from score_amd.native import graph_arrays
from score_amd.solver import ConicSolver, host_counters

sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
arrs = [graph_arrays(make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=4000 + t)) for t in range(64)]
tick = os.sysconf("SC_CLK_TCK")

def threads():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            s = open(f"/proc/self/task/{tid}/stat").read()
            comm = s[s.index("(") + 1:s.rindex(")")]
            f = s[s.rindex(")") + 2:].split()
            out[int(tid)] = (comm, (int(f[11]) + int(f[12])) / tick, int(f[11]) / tick, int(f[12]) / tick)
        except OSError:
            pass
    return out

def one(k):
    sv = ConicSolver.from_graphs(arrs[16 * k:16 * k + 16], 0, dict(device=0))
    try:
        infos, ests = sv.solve_estimates()
    finally:
        sv.close()
    return sum(i["status"] == 1 for i in infos)

with ThreadPoolExecutor(max_workers=4) as pool:
    for _ in range(2):
        list(pool.map(one, range(4)))
    a, w0, t0 = threads(), host_counters(), time.perf_counter()
    for _ in range(sweeps):
        ok = sum(pool.map(one, range(4)))
    dt = time.perf_counter() - t0
    b, w1 = threads(), host_counters()
n = sweeps * 64
print(f"{sweeps} sweeps of 64 fresh graphs: {1e3 * dt / sweeps:.2f} ms per sweep, {n / dt:.0f} problems/s, solved {ok}/64 in the last")
rows = sorted(((b[t][1] - a.get(t, (0, 0, 0, 0))[1], b[t][2] - a.get(t, (0, 0, 0, 0))[2], b[t][3] - a.get(t, (0, 0, 0, 0))[3], b[t][0], t) for t in b), reverse=True)
tot = sum(r[0] for r in rows)
print(f"host CPU {1e3 * tot / n:.3f} ms per problem ({tot / dt:.2f} CPUs busy); library waits: spinning {(w1['spin_ms'] - w0['spin_ms']) / n:.3f} ms, asleep {(w1['sleep_ms'] - w0['sleep_ms']) / n:.3f} ms per problem")
for cpu, ut, st, comm, tid in rows[:14]:
    print(f"  {comm:18s} tid {tid:7d}: {1e3 * cpu / n:6.3f} ms per problem (user {1e3 * ut / n:.3f}, sys {1e3 * st / n:.3f})")
