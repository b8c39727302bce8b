"""One default solve from a rocprofv3 kernel trace (rocpd sqlite): kernels, busy time, and the GAPS between consecutive
kernels -- the host waits of the Newton iterations show up as the gaps above ~8 us.  Takes the LAST solve of the trace
(delimited by the reset's fill).  python r05_solve_gaps.py results.db"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
# solves start with the reset: a fillBufferAligned followed (soon) by k_refresh_u
starts = [i for i, r in enumerate(rows) if "k_refresh_u" in r[0]]
# the last two refresh_u launches belong to ... take the kernels after the second-to-last reset up to the end
if len(starts) >= 2:
    i0 = starts[-2] - 1
    # find the end: the next reset
    i1 = starts[-1] - 1
else:
    i0, i1 = 0, len(rows)
sel = rows[i0:i1]
t0, t1 = sel[0][1], sel[-1][2]
busy = sum(e - s for _, s, e in sel)
print(f"one default solve: {len(sel)} kernels over {1e-3*(t1-t0):.1f} us; kernel time {1e-3*busy:.1f} us ({100*busy/(t1-t0):.1f} % busy)")
wait = sum(e - s for n, s, e in sel if "k_fetch_wait" in n)
if wait:
    print(f"   of which k_fetch_wait (one workgroup polling a host-mapped flag: the host's decision) {1e-3*wait:.1f} us; without it {100*(busy-wait)/(t1-t0):.1f} % busy")
gaps = []
for (n0, s0, e0), (n1, s1, e1) in zip(sel[:-1], sel[1:]):
    gaps.append((s1 - e0, n0.split("(")[0][-40:], n1.split("(")[0][-40:]))
big = [g for g in gaps if g[0] > 8000]
small = [g for g in gaps if g[0] <= 8000]
print(f"gaps: {len(small)} below 8 us, sum {1e-3*sum(g[0] for g in small):.1f} us (mean {1e-3*sum(g[0] for g in small)/max(1,len(small)):.2f}); {len(big)} above, sum {1e-3*sum(g[0] for g in big):.1f} us")
for g in sorted(big, key=lambda x: -x[0])[:16]:
    print(f"   {1e-3*g[0]:7.1f} us between {g[1]} and {g[2]}")
agg = {}
for n, s, e in sel:
    k = n.split("(")[0][:64]
    a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"  {1e-3*t:8.1f} us {100*t/busy:5.1f} % {c:4d} x {1e-3*t/c:7.2f}  {k}")
