"""GPU idle time between consecutive kernels of a rocprofv3 kernel trace (``*_kernel_trace.csv``), attributed to the kernel that
runs AFTER the gap: python scripts/gap_analysis.py trace.csv [last_seconds]   (only the last N seconds of the trace are analysed)."""
import collections
import csv
import sys

path = sys.argv[1]
last = float(sys.argv[2]) if len(sys.argv) > 2 else None
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
if last is not None:
    t_end = rows[-1][1]
    rows = [r for r in rows if r[0] >= t_end - last * 1e9]


def short(n):
    n = n.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    return n.split('(')[0][:60]


busy = 0
pairs = collections.defaultdict(lambda: [0, 0.])
prev_name = None
gaps = collections.defaultdict(lambda: [0, 0.])
big = collections.defaultdict(lambda: [0, 0.])
prev_end = rows[0][0]
for s, e, n in rows:
    if s > prev_end:
        g = (s - prev_end) / 1e3
        key = short(n)
        gaps[key][0] += 1
        gaps[key][1] += g
        if g > 15.:
            big[key][0] += 1
            big[key][1] += g
        pk = (short(prev_name) if prev_name else '-', key)
        pairs[pk][0] += 1
        pairs[pk][1] += g
    busy += max(0, e - max(s, prev_end))
    if e >= prev_end:
        prev_name = n
    prev_end = max(prev_end, e)
span = (rows[-1][1] - rows[0][0]) / 1e9
print("span %.3f s, kernels %d, busy %.3f s (%.1f %%), idle %.3f s" % (span, len(rows), busy / 1e9, 100 * busy / 1e9 / span, span - busy / 1e9))
print("idle time by the kernel that follows the gap (all gaps | gaps > 15 us):")
for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print("  %-62s %7d gaps %9.1f ms (avg %6.1f us) | %6d long %9.1f ms" % (k, c, t / 1e3, t / c, big[k][0], big[k][1] / 1e3))
print("idle time by (kernel before the gap -> kernel after it):")
for k, (c, t) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:40]:
    print("  %-44s -> %-44s %6d gaps %8.1f ms (avg %6.1f us)" % (k[0][:44], k[1][:44], c, t / 1e3, t / c))
