"""Time SO(d) rounding of 20 000 blocks: NumPy (closed form d=2 / batched SVD d=3) vs score_round_to_so
on the device (pinned staging, kernel reads/writes across the link)."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from score_amd.rounding import round_to_special_orthogonal
from score_amd.solver import load_library
lib = load_library()
rng = np.random.default_rng(0)
for d in (2, 3):
    q, _ = np.linalg.qr(rng.normal(size=(20000, d, d)))
    M = q + 1e-3 * rng.normal(size=q.shape)
    for name, kw in (("numpy", {}), ("device", dict(lib=lib))):
        ts = []
        for _ in range(6):
            t = time.perf_counter(); R = round_to_special_orthogonal(M, **kw); ts.append(time.perf_counter() - t)
        print(f"d={d} {name:6s} first {ts[0]*1e3:8.3f} ms  best {min(ts)*1e3:8.3f} ms")
