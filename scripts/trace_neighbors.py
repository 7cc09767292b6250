"""Durations of the kernels around every launch of a marker kernel in a rocprofv3 kernel trace (last N seconds):
python scripts/trace_neighbors.py trace.csv seconds marker_substring [before after]"""
import collections
import csv
import sys

path, last, marker = sys.argv[1], float(sys.argv[2]), sys.argv[3]
nb, na = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (2, 3)
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
t_end = rows[-1][1]
rows = [r for r in rows if r[0] >= t_end - last * 1e9]


def short(n):
    n = n.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    return n.split('(')[0][:50]


agg = collections.defaultdict(lambda: [0, 0., 0.])
for i, (s, e, n) in enumerate(rows):
    if marker not in n:
        continue
    for off in range(-nb, na + 1):
        j = i + off
        if 0 <= j < len(rows):
            a = agg[(off, short(rows[j][2]))]
            a[0] += 1
            a[1] += rows[j][1] - rows[j][0]
            if j > 0:
                a[2] += rows[j][0] - rows[j - 1][1]
print("offset kernel calls avg_us avg_gap_before_us")
for (off, k), (c, t, g) in sorted(agg.items()):
    if c >= 20:
        print("%+d %-50s %6d %8.1f %8.1f" % (off, k, c, t / 1e3 / c, g / 1e3 / c))
