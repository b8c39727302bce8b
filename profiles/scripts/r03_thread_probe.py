"""Do plain C calls from the threads of ONE process scale on this host?  memset of a private 16 MB buffer per call
(libc through ctypes, GIL released).  Measured on the MI355X host: 52 GB/s on one thread, 135 GB/s from 4 threads on --
the threads a process wakes stay on one L3 domain (one CCD, one link to memory)."""
import ctypes as C, time
from concurrent.futures import ThreadPoolExecutor
libc = C.CDLL("libc.so.6")
libc.memset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]; libc.memset.restype = C.c_void_p
bufs = [C.create_string_buffer(16 << 20) for _ in range(16)]
def call(i):
    t = time.perf_counter(); libc.memset(bufs[i % 16], i & 255, 16 << 20); return time.perf_counter() - t
for i in range(16): call(i)
for k in (1, 2, 4, 8, 16):
    best = None
    for _ in range(4):
        t = time.perf_counter()
        with ThreadPoolExecutor(max_workers=k) as pool: each = list(pool.map(call, range(128)))
        wall = time.perf_counter() - t
        if best is None or wall < best[0]: best = (wall, each)
    print(f"128 memsets of 16 MB on {k} threads: wall {1e3*best[0]:.1f} ms ({128*16/1024/best[0]:.0f} GB/s), mean call {1e3*sum(best[1])/128:.3f} ms", flush=True)
