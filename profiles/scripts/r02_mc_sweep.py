"""BASELINE configs[4] on one GPU: problems/s against (trials per handle, host threads)."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from concurrent.futures import ThreadPoolExecutor
models = None
import sys
CONFIGS = ((16, 4), (8, 8), (16, 8), (11, 6), (22, 3), (32, 2), (64, 1), (4, 16)) if len(sys.argv) < 2 else tuple(tuple(int(v) for v in a.split('x')) for a in sys.argv[1:])
for per, thr in CONFIGS:
    args = bench.parse_args(["--mc-batch", str(per), "--mc-threads", str(thr)])
    mc = bench.MonteCarlo(args, range(64), 0, None)
    with ThreadPoolExecutor(max_workers=mc.threads) as pool:
        mc.sweep(pool)
        t0 = time.perf_counter()
        for _ in range(8):
            last = mc.sweep(pool)
        dt = time.perf_counter() - t0
    print(f"groups {mc.group_sizes} threads {thr}: {64*8/dt:.0f} problems/s, {1e3*dt/8:.2f} ms per sweep, solved {sum(o.solved for o in last)}", flush=True)
    mc.close()
