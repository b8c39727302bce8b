"""GPU busy fraction and per-kernel time of one window of a rocprofv3 kernel trace (rocpd sqlite output).
python r05_trace_busy.py results.db [t0_ns t1_ns | last <ms>]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
if len(sys.argv) > 3 and sys.argv[2] == "last":  # the last <ms> of the trace (the trace's clock is not the wall clock)
    t1 = max(r[2] for r in rows); t0 = t1 - int(float(sys.argv[3]) * 1e6)
elif len(sys.argv) > 3:
    t0, t1 = int(sys.argv[2]), int(sys.argv[3])
else:
    t0, t1 = rows[0][1], rows[-1][2]
sel = [(n, max(s, t0), min(e, t1)) for n, s, e in rows if e > t0 and s < t1]


def union(iv):  # total length of the union of intervals
    busy, cs, cur_e = 0, None, None
    for n, s, e in sorted(iv, key=lambda r: r[1]):
        if cs is None: cs, cur_e = s, e
        elif s <= cur_e: cur_e = max(cur_e, e)
        else: busy += cur_e - cs; cs, cur_e = s, e
    if cs is not None: busy += cur_e - cs
    return busy


busy = union(sel)
tot = sum(e - s for _, s, e in sel)
print(f"window {1e-6*(t1-t0):.2f} ms: {len(sel)} kernels, busy {1e-6*busy:.2f} ms = {100*busy/(t1-t0):.1f} %, kernel time {1e-6*tot:.2f} ms (overlap {tot/max(1,busy):.2f})")
# k_fetch_wait is a kernel that WAITS for the host (one workgroup polling a flag): counted as idle here
work = [r for r in sel if "k_fetch_wait" not in r[0]]
if len(work) != len(sel):
    b2 = union(work)
    print(f"  without the kernels that wait for the host (k_fetch_wait, {len(sel) - len(work)} x): busy {1e-6*b2:.2f} ms = {100*b2/(t1-t0):.1f} %")
agg = {}
for n, s, e in sel:
    k = n.split("(")[0][:70]
    a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"  {1e-3*t:9.1f} us {100*t/tot:5.1f} % {c:5d} x {1e-3*t/c:7.2f}  {k}")
