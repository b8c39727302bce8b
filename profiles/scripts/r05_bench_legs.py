"""The bench's Monte-Carlo legs in isolation and in sequence (are they slowed by what ran before them in the process?).
python profiles/scripts/r05_bench_legs.py e2e | c5 | c5_e2e | head_c5_e2e"""
import json, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import bench
which = sys.argv[1] if len(sys.argv) > 1 else "e2e"
args = bench.parse_args([])
D = bench.Dist(args)
if which.startswith("pre"):   # preludes, one ingredient at a time
    from score_amd.manhattan import make_manhattan
    from score_amd.native import assemble_native, graph_arrays
    from score_amd.solver import ConicSolver
    fg = make_manhattan(n_robots=20, n_poses=1000, n_beacons=4, seed=3000)
    if "asm" in which: m = assemble_native(fg, "SOCP")                       # host assembler only (thread teams)
    if "graphh" in which:                                                    # a headline handle from the graph, solved
        s = ConicSolver.from_graphs([graph_arrays(fg)], 0, dict(device=0)); [s.solve() for _ in range(5)]; s.close()
    if "qph" in which:                                                       # a headline handle from the host-assembled program
        m = assemble_native(fg, "SOCP"); s = ConicSolver([m.qp], dict(device=0)); [s.solve() for _ in range(5)]; s.close()
    if "smallh" in which:                                                    # a 4-robot handle from the graph, solved
        s = ConicSolver.from_graphs([graph_arrays(make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=1))], 0, dict(device=0)); [s.solve() for _ in range(5)]; s.close()
    if "admmh" in which:                                                     # a headline handle, ADMM only
        s = ConicSolver.from_graphs([graph_arrays(fg)], 0, dict(device=0, polish=0, max_iters=200)); [s.solve() for _ in range(2)]; s.close()
    if "createh" in which:                                                   # a headline handle, created and destroyed, never solved
        ConicSolver.from_graphs([graph_arrays(fg)], 0, dict(device=0)).close()
    if "nopolh" in which:                                                    # created without the polish, never solved
        ConicSolver.from_graphs([graph_arrays(fg)], 0, dict(device=0, polish=0)).close()
    if "nograph" in which:                                                   # a headline handle solved WITHOUT launch graphs
        s = ConicSolver.from_graphs([graph_arrays(fg)], 0, dict(device=0, use_graph=0)); [s.solve() for _ in range(5)]; s.close()
    if "thr" in which:                                                       # ... solved from another thread
        import threading
        def run():
            s = ConicSolver.from_graphs([graph_arrays(fg)], 0, dict(device=0)); [s.solve() for _ in range(5)]; s.close()
        t = threading.Thread(target=run); t.start(); t.join()
    if "trim" in which:
        from score_amd.solver import trim_caches
        print("trimmed", trim_caches())
if which.startswith("head"):
    from score_amd.solver import ConicSolver
    models = bench.make_headline(args, 0, 1)
    s = ConicSolver([m.qp for m in models], dict(device=0)); [s.solve() for _ in range(5)]
    if "probe1" in which: s.time_iteration(warmup=10, iters=40, dispatch=True)   # (event-bound launches on a single-problem handle)
    s.close()
    if "nobatch" not in which:
        bm = models + bench.make_headline(args, 1, 15)
        s = ConicSolver([m.qp for m in bm], dict(device=0, polish=0)); s.time_iteration(warmup=10, iters=40, dispatch=("noev" not in which)); s.close()
if "c5" in which:
    c5 = bench.config5_leg(args, D)
    print("config5", round(c5["problems_per_sec"]), "fresh", round(c5["fresh_graphs_problems_per_sec"]), [round(t, 1) for t in c5["fresh_graphs"]["ms_per_sweep_rank0"]], flush=True)
if "e2e" in which:
    e = bench.end_to_end(args, 0)
    print("e2e from arrays", round(e["config4_end_to_end_from_arrays_problems_per_sec"]), [round(t, 1) for t in e["config4_sweeps_ms"]["from_arrays"]],
          "objects", round(e["config4_end_to_end_problems_per_sec"]), "headline", round(e["headline_solve_score_ms"], 2), "create", round(e["score_create_ms_best"], 2), flush=True)
