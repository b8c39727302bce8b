"""Create / solve / destroy in a loop (single handles and 4 concurrent lock-step groups): host RSS and device memory must level off."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import resource
import torch
from score_amd.manhattan import make_manhattan
from score_amd.native import ArrayGraph, graph_arrays
from score_amd.solve_score import solve_score, solve_score_batch
def rss(): return int(open("/proc/self/statm").read().split()[1]) * 4096 / 2**20
def dev(): f, t = torch.cuda.mem_get_info(0); return (t - f) / 2**20
big = ArrayGraph(graph_arrays(make_manhattan(n_robots=20, n_poses=1000, n_beacons=4, seed=3000)))
small = [ArrayGraph(graph_arrays(make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=5000 + t))) for t in range(32)]
import ctypes
libc = ctypes.CDLL('libc.so.6')
class MI(ctypes.Structure):
    _fields_ = [(n, ctypes.c_size_t) for n in ('arena','ordblks','smblks','hblks','hblkhd','usmblks','fsmblks','uordblks','fordblks','keepcost')]
libc.mallinfo2.restype = MI
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for rnd in range(ROUNDS):
    for _ in range(25): solve_score(big, "SOCP")
    for _ in range(5): solve_score_batch(small, "SOCP", workers=4)
    nthreads = len(os.listdir("/proc/self/task"))
    mi = libc.mallinfo2()
    print(f"round {rnd}: host RSS {rss():.0f} MB (main arena {mi.arena/2**20:.0f} MB, in use {mi.uordblks/2**20:.0f} MB, free {mi.fordblks/2**20:.0f} MB, mmapped {mi.hblkhd/2**20:.0f} MB), "
          f"device in use {dev():.0f} MB, threads {nthreads}", flush=True)
from score_amd.solver import trim_caches
freed = trim_caches()
print(f"score_trim_caches(): {freed/2**20:.0f} MB of device / pinned blocks released; host RSS {rss():.0f} MB, device in use {dev():.0f} MB", flush=True)
