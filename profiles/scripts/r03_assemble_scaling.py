import os, sys, time, resource, ctypes as C; sys.path.insert(0, os.getcwd())
from concurrent.futures import ThreadPoolExecutor
if os.environ.get("NODE") is not None:
    def cpus(node):
        out = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-"); out.update(range(int(a), int(b or a) + 1))
        return out
    os.sched_setaffinity(0, cpus(int(os.environ["NODE"])) & os.sched_getaffinity(0))
    print("pinned to node", os.environ["NODE"], len(os.sched_getaffinity(0)), "cpus")
from score_amd.manhattan import make_manhattan
from score_amd.native import graph_arrays, score_graph_struct, _bind
from score_amd.solver import load_library
lib = load_library(None); _bind(lib)
fgs = [make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=5000 + t) for t in range(64)]
arrs = [graph_arrays(fg) for fg in fgs]
gs = [score_graph_struct(a, 0) for a in arrs]
import threading, itertools, glob
def l3_domains():
    doms = {}
    for c in sorted(os.sched_getaffinity(0)):
        try: key = open(f"/sys/devices/system/cpu/cpu{c}/cache/index3/shared_cpu_list").read().strip()
        except OSError: return []
        doms.setdefault(key, set()).add(c)
    return list(doms.values())
DOMS = l3_domains() if os.environ.get("SPREAD") else []
_next = itertools.count()
def spread():
    if DOMS: os.sched_setaffinity(threading.get_native_id(), DOMS[(next(_next) * (1 if os.environ.get("SPREAD") == "1" else 8)) % len(DOMS)])
def call(i):
    h = C.c_void_p()
    t = time.perf_counter(); rc = lib.score_assemble(C.byref(gs[i]), C.byref(h)); dt = time.perf_counter() - t
    t2 = time.perf_counter(); lib.score_assembled_free(h); df = time.perf_counter() - t2
    return dt, df
for i in range(8): call(i)
for k in (1, 2, 4, 8, 16, 4, 1):
    best = None
    for _ in range(4):
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        t = time.perf_counter()
        with ThreadPoolExecutor(max_workers=k, initializer=spread) as pool: each = list(pool.map(call, range(64)))
        wall = time.perf_counter() - t
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        if best is None or wall < best[0]: best = (wall, each, ru1.ru_minflt - ru0.ru_minflt, ru1.ru_nvcsw - ru0.ru_nvcsw, ru1.ru_nivcsw - ru0.ru_nivcsw, ru1.ru_stime - ru0.ru_stime, ru1.ru_utime - ru0.ru_utime)
    print(f"64 C calls on {k} threads: wall {1e3*best[0]:.1f} ms, mean assemble {1e3*sum(e[0] for e in best[1])/64:.2f} ms, mean free {1e3*sum(e[1] for e in best[1])/64:.3f} ms; page faults {best[2]}, voluntary / involuntary switches {best[3]} / {best[4]}, sys {1e3*best[5]:.1f} ms user {1e3*best[6]:.1f} ms", flush=True)
