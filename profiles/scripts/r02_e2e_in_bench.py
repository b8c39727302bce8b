"""Why solve_score() end to end is slower inside bench.py (117 ms) than alone (78 ms): candidates are
Python's generational GC walking the bench's live graphs, torch's threads, cached blocks."""
import gc, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
if "--torch" in sys.argv:  # torch first: it brings its own HIP runtime, which must be the one that opens the device
    import torch
    torch.cuda.init(); _x = torch.zeros(10, device="cuda:0"); torch.cuda.synchronize()
from score_amd.manhattan import make_manhattan
from score_amd.solve_score import solve_score, solve_score_batch

def timed(tag, fg, n=5):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); r = solve_score(fg, "SOCP"); ts.append(time.perf_counter() - t)
    print(f"{tag:40s} mean {1e3*sum(ts)/n:7.1f} ms  min {1e3*min(ts):7.1f} ms  solved {r.solved}", flush=True)

fg = make_manhattan(n_robots=20, n_poses=1000, n_beacons=4, seed=3000)
solve_score(fg, "SOCP")
timed("alone" + (" (torch + cuda initialised)" if "--torch" in sys.argv else ""), fg)
trials = [make_manhattan(n_robots=4, n_poses=500, n_beacons=2, seed=100 + t) for t in range(64)]
timed("with 64 live trial graphs", fg)
solve_score_batch(trials, "SOCP", workers=4)
timed("after a 64-trial batch (4 threads)", fg)
gc.collect(); gc.freeze()
timed("after gc.freeze()", fg)
gc.disable()
timed("gc disabled", fg)
