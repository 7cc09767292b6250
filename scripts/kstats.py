"""Kernel statistics from a rocprofv3 result database (rocpd sqlite): name, calls, avg / min us, total ms."""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'info_kernel_symbol' in t][0]
q = ("select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), sum(d.end-d.start) from %s d join %s s "
     "on d.kernel_id = s.id group by s.kernel_name order by 5 desc" % (kd, ks))
tot = 0
rows = list(cur.execute(q))
for r in rows:
    tot += r[4]
print("kernel,calls,avg_us,min_us,total_ms,percent")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%s,%d,%.2f,%.2f,%.3f,%.1f" % (r[0][:90].replace(',', ';'), r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e6, 100. * r[4] / tot))
