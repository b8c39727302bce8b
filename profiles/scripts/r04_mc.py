"""BASELINE configs[4] on one GPU as the bench leg runs it (64 trials, four lock-step handles of 16, result gather per sweep):
    python profiles/scripts/r04_mc.py [sweeps=10]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
args = bench.parse_args([])
D = bench.Dist(args)
sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
r = bench.config5_leg(args, D, 64, sweeps)
print({k: r[k] for k in ("problems_per_sec", "ms_per_sweep", "solved_last_sweep")})
