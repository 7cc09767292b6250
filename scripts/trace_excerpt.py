"""Raw excerpt of a rocprofv3 kernel trace: n kernels starting `seconds_before_end` before the end (name, duration, gap before).
python scripts/trace_excerpt.py trace.csv seconds_before_end n"""
import csv
import sys
path, back, n = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Grid_Size', r.get('Grid_Size_X', ''))))
rows.sort()
t0 = rows[-1][1] - back * 1e9
i0 = next(i for i, r in enumerate(rows) if r[0] >= t0)
prev = rows[i0 - 1][1] if i0 else rows[0][0]
for s, e, nme, g in rows[i0:i0 + n]:
    nme = nme.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '').split('(')[0][:48]
    print("%8.1f us gap %7.1f  %-48s grid %s" % ((e - s) / 1e3, (s - prev) / 1e3, nme, g))
    prev = e
