"""Per-call time of score_assemble when k PROCESSES run it at once (against k threads of one process:
r03_assemble_scaling.py).  Measured on the MI355X host: 1.22 ms per call at 1, 2, 4 and 8 processes; in one process the
call takes 1.3 ms on one thread, 2.0 on two, 3.9 on sixteen."""
import os, sys, time, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import ctypes as C
    sys.path.insert(0, os.getcwd())
    from score_amd.manhattan import make_manhattan
    from score_amd.native import graph_arrays, score_graph_struct, _bind
    from score_amd.solver import load_library
    lib = load_library(None); _bind(lib)
    fgs = [make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=5000 + t) for t in range(4)]
    arrs = [graph_arrays(fg) for fg in fgs]
    gs = [score_graph_struct(a, 0) for a in arrs]
    def call(i):
        h = C.c_void_p(); t = time.perf_counter(); rc = lib.score_assemble(C.byref(gs[i % 4]), C.byref(h)); dt = time.perf_counter() - t
        assert rc == 0
        lib.score_assembled_free(h); return dt
    for i in range(8): call(i)
    start = float(sys.argv[2])
    while time.time() < start: pass
    ts = [call(i) for i in range(200)]
    print(f"{1e3*sum(ts)/len(ts):.3f}")
    sys.exit(0)
for k in (1, 2, 4, 8):
    start = time.time() + 6.0
    ps = [subprocess.Popen([sys.executable, __file__, "child", str(start)], stdout=subprocess.PIPE, text=True, env=dict(os.environ, SCORE_HOST_THREADS="1")) for _ in range(k)]
    outs = [p.communicate()[0].strip() for p in ps]
    print(f"{k} processes: mean call ms per process {outs}", flush=True)
