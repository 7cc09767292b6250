"""Per-kernel busy time inside the LAST N seconds of a rocprofv3 kernel trace (``*_kernel_trace.csv``): the timed sweeps of a bench
run without the ramp.  python scripts/trace_window.py trace.csv seconds [top]"""
import collections
import csv
import sys

path, last = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
t_end = rows[-1][1]
rows = [r for r in rows if r[0] >= t_end - last * 1e9]


def short(n):
    n = n.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    return n.split('(')[0][:70]


agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    a = agg[short(n)]
    a[0] += 1
    a[1] += e - s
span = (rows[-1][1] - rows[0][0]) / 1e9
busy = sum(a[1] for a in agg.values()) / 1e9
print("window %.3f s, %d kernels, busy %.3f s (%.1f %%)" % (span, len(rows), busy, 100 * busy / span))
print("%-70s %8s %10s %9s %7s" % ("kernel", "calls", "total ms", "avg us", "% span"))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%-70s %8d %10.2f %9.2f %7.2f" % (k, c, t / 1e6, t / 1e3 / c, 100 * t / 1e9 / span))
