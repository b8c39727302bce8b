"""Does the placement of host memory bound score_create / the end-to-end sweep?  Prints the NUMA layout of the box, then
times the headline create and a from-arrays sweep of configs[4]; `interleave` as first argument sets MPOL_INTERLEAVE over
all nodes for the process (set_mempolicy, before anything is allocated).
python profiles/scripts/r04_numa.py [default|interleave|local]"""
import ctypes, glob, os, sys, time
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
nodes = sorted(int(p.rsplit("node", 1)[1]) for p in glob.glob("/sys/devices/system/node/node[0-9]*"))
if mode == "default":
    print("nodes:", nodes)
    for nd in nodes[:16]:
        try:
            cl = open(f"/sys/devices/system/node/node{nd}/cpulist").read().strip()
            mem = [l for l in open(f"/sys/devices/system/node/node{nd}/meminfo") if "MemTotal" in l or "MemFree" in l]
            print(f"  node {nd}: cpus {cl}; " + "; ".join(" ".join(l.split()[2:]) for l in mem))
        except OSError as e:
            print("  node", nd, e)
    print("affinity:", len(os.sched_getaffinity(0)), "cpus", sorted(os.sched_getaffinity(0))[:8], "...")
    try: print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip(), "cpuset:", open("/sys/fs/cgroup/cpuset.cpus.effective").read().strip(), "mems:", open("/sys/fs/cgroup/cpuset.mems.effective").read().strip())
    except OSError as e: print(e)
if mode == "interleave" and len(nodes) > 1:
    libc = ctypes.CDLL(None, use_errno=True)
    mask = ctypes.c_ulong(sum(1 << n for n in nodes))
    rc = libc.syscall(238, 3, ctypes.byref(mask), ctypes.c_ulong(max(nodes) + 2))  # set_mempolicy(MPOL_INTERLEAVE)
    print("set_mempolicy(MPOL_INTERLEAVE) rc", rc, "errno", ctypes.get_errno())
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from score_amd.manhattan import make_manhattan
from score_amd.native import ArrayGraph, assemble_native, graph_arrays
from score_amd.solver import ConicSolver
import score_amd.solve_score as S
fg = make_manhattan(n_robots=20, n_poses=1000, n_beacons=4, seed=3000)
arr = graph_arrays(fg)
m = assemble_native(fg, "SOCP", arrays=arr)
ConicSolver([m.qp], {}).close()
cr, asm = [], []
for i in range(15):
    t = time.perf_counter(); m = assemble_native(fg, "SOCP", arrays=arr); asm.append(time.perf_counter() - t)
    t = time.perf_counter(); s = ConicSolver([m.qp], {}); cr.append(time.perf_counter() - t); s.close()
print(f"[{mode}] headline: assemble min {1e3*min(asm):.2f} ms, create min {1e3*min(cr):.2f} median {1e3*sorted(cr)[7]:.2f} ms")
trials = [make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=5000 + t) for t in range(64)]
flat = [ArrayGraph(graph_arrays(f)) for f in trials]
for w in (8, 4):
    ts = []
    for rep in range(5):
        t0 = time.perf_counter(); rs = S.solve_score_batch(flat, "SOCP", solver_settings=dict(device=0), workers=w); ts.append(time.perf_counter() - t0)
    print(f"[{mode}] 64 trials from arrays, workers {w}: best {64/min(ts[1:]):.0f} graphs/s, median {64/sorted(ts[1:])[2]:.0f}")
