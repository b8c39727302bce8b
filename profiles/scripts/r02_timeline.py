"""Summarise a rocprofv3 kernel trace csv of r02_trace_polish.py: for the LAST solve, kernel time by
name, number of launches, idle gaps between consecutive GPU activities."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# find solves: split on gaps > 300 us
groups, cur = [], [ev[0]]
for e in ev[1:]:
    if e[0] - cur[-1][1] > 300_000:
        groups.append(cur); cur = [e]
    else:
        cur.append(e)
groups.append(cur)
g = groups[-1]
tot = (g[-1][1] - g[0][0]) / 1e3
busy = sum(e[1] - e[0] for e in g) / 1e3
print(f"last solve: {len(g)} launches, span {tot:.1f} us, kernel time {busy:.1f} us, idle {tot-busy:.1f} us")
by = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in g:
    k = n.split("(")[0].replace("void score::", "").replace("score::", "")
    by[k][0] += 1; by[k][1] += (e - s) / 1e3
for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:40s} {c:5d} launches {t:9.1f} us  avg {t/c:7.2f}")
gaps = sorted(((g[i+1][0] - g[i][1]) / 1e3, g[i][2].split('(')[0][-30:], g[i+1][2].split('(')[0][-30:]) for i in range(len(g)-1))
print("largest gaps (us, after, before):")
for x in gaps[-25:]:
    print("   %.1f  %s -> %s" % x)
print("gap histogram: <1us %d, 1-3 %d, 3-10 %d, 10-30 %d, >30 %d" % (
    sum(x[0] < 1 for x in gaps), sum(1 <= x[0] < 3 for x in gaps), sum(3 <= x[0] < 10 for x in gaps),
    sum(10 <= x[0] < 30 for x in gaps), sum(x[0] >= 30 for x in gaps)))
