"""Memory copies of a rocprofv3 --memory-copy-trace run (``*_memory_copy_trace.csv``): counts and bytes by direction and size class
over the last N seconds: python scripts/copy_analysis.py trace.csv [last_seconds]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
last = float(sys.argv[2]) if len(sys.argv) > 2 else None
t_end = max(int(r['End_Timestamp']) for r in rows)
if last is not None:
    rows = [r for r in rows if int(r['Start_Timestamp']) >= t_end - last * 1e9]
print(rows[0].keys())
agg = collections.defaultdict(lambda: [0, 0, 0.])
for r in rows:
    size = int(r.get('Size', r.get('Bytes', 0)) or 0)
    cls = '<=64B' if size <= 64 else '<=4KB' if size <= 4096 else '<=256KB' if size <= (1 << 18) else '<=16MB' if size <= (1 << 24) else '>16MB'
    k = (r.get('Direction', r.get('Kind', '?')), cls)
    agg[k][0] += 1
    agg[k][1] += size
    agg[k][2] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, (c, b, t) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("%-28s %-9s %7d copies %10.2f MB %9.1f ms" % (k[0], k[1], c, b / 1e6, t / 1e3))
