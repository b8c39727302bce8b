"""Does the process context slow the fresh-graph sweeps down (bench.py: 40-46 ms per sweep of 8 x 8, stand-alone 32 ms)?
Variants: plain; torch imported and its GPU context initialised; after a host-assembled create (host thread teams exist).
python profiles/scripts/r05_fresh_ctx.py <variant>"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
variant = sys.argv[1] if len(sys.argv) > 1 else "plain"
if variant in ("torch", "torch_teams"):
    import torch
    torch.cuda.synchronize()
from score_amd.manhattan import make_manhattan
from score_amd.native import assemble_native, graph_arrays
from score_amd.solver import ConicSolver
if variant in ("teams", "torch_teams"):
    m = assemble_native(make_manhattan(n_robots=20, n_poses=1000, n_beacons=4, seed=1), "SOCP")
    os.environ["SCORE_HOST_SETUP"] = "1"; ConicSolver([m.qp], {}).close(); del os.environ["SCORE_HOST_SETUP"]
N = 64
arrs = [graph_arrays(make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=4000 + t)) for t in range(N)]
def one(idx):
    s = ConicSolver.from_graphs([arrs[i] for i in idx], 0, {})
    try:
        return s.solve()
    finally:
        s.close()
groups = [list(range(i, i + 8)) for i in range(0, N, 8)]
with ThreadPoolExecutor(8) as pool:
    for _ in range(2): list(pool.map(one, groups))
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); list(pool.map(one, groups)); ts.append(1e3 * (time.perf_counter() - t0))
print(f"{variant:12s} sweeps of 8 x 8: " + " ".join(f"{t:5.1f}" for t in ts) + f"  median {sorted(ts)[4]:.1f} ms", flush=True)
