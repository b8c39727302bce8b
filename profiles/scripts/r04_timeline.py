"""Per-workgroup timeline of one ADMM iteration in the loop (device wall clock stamps of every workgroup, SCORE_DUMP_STAMPS):
for each of the six kernels the entry of its first and the exit of its last workgroup, the median / longest workgroup, and
the workgroups that leave last.    python profiles/scripts/r04_timeline.py [batch=1]"""
import os, re, sys, subprocess, collections; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if os.environ.get("SCORE_DUMP_STAMPS") != "1":
    env = dict(os.environ, SCORE_DUMP_STAMPS="1")
    out = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, capture_output=True, text=True)
    rows = collections.defaultdict(list)
    for k, b, a0, a1 in re.findall(r"STAMP (\d+) (\d+) ([0-9.]+) ([0-9.]+)", out.stdout): rows[int(k)].append((int(b), float(a0), float(a1)))
    print("\n".join(ln for ln in out.stdout.splitlines() if "STAMP" not in ln))
    names = ("rhs", "prec_init", "kp", "prec_step", "kpb", "cone")
    for k in sorted(rows):
        r = rows[k]; t0 = min(x[1] for x in r); t1 = max(x[2] for x in r)
        durs = sorted(x[2] - x[1] for x in r)
        last_in = max(x[1] for x in r)
        slow = sorted(r, key=lambda x: -x[2])[:5]
        print(f"{names[k]:9s} {len(r):4d} wgs: first in {t0:7.2f}, last in {last_in:7.2f}, last out {t1:7.2f} (span {t1-t0:5.2f}); wg time median {durs[len(durs)//2]:.2f} max {durs[-1]:.2f};"
              f" last out: " + ", ".join(f"wg{b}[{a0-t0:.2f}-{a1-t0:.2f}]" for b, a0, a1 in slow))
    print(out.stderr[-2000:])
    sys.exit(0)
from score_amd.manhattan import make_manhattan
from score_amd.native import assemble_native
from score_amd.solver import ConicSolver
B = 1
for a in sys.argv[1:]:
    if a.startswith("batch="): B = int(a[6:])
models = [assemble_native(make_manhattan(n_robots=20, n_poses=1000, n_beacons=4, seed=3000 + j), "SOCP") for j in range(B)]
s = ConicSolver([m.qp for m in models], dict(polish=0, adaptive_cg=0))
dev = s.time_iteration(warmup=20, iters=20)
print("device clock:", " ".join(f"{k} {v:.2f}" for k, v in dev.items()))
s.close()
