"""200 refinement handles created, run and destroyed: device memory in use must not grow (the buffers live in the
handle's arena and go back to the block cache with it)."""
import ctypes, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from score_amd.manhattan import make_manhattan
from score_amd.refine import refine_estimate
from test_refine import _noisy_truth
hip = ctypes.CDLL("libamdhip64.so")
def used_mb():
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
    return (total.value - free.value) / 2**20
fg = make_manhattan(n_robots=4, n_poses=500, n_beacons=3, seed=11)
res = _noisy_truth(fg)
refine_estimate(fg, res)
m0 = used_mb(); t0 = time.perf_counter()
for i in range(200):
    out, info = refine_estimate(fg, res)
    if i % 50 == 49:
        print(f"{i+1} handles: device memory in use {used_mb():.0f} MB (start {m0:.0f} MB), {1e3*(time.perf_counter()-t0)/(i+1):.1f} ms per refine_estimate", flush=True)
assert used_mb() - m0 < 64, "device memory grew"
print("ok")
