"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, dispatch count and mean counter value."""
import csv
import sys
from collections import defaultdict
path = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
with open(path) as f:
    for row in csv.DictReader(f):
        acc[row['Kernel_Name'][:90]][row['Counter_Name']].append(float(row['Counter_Value']))
print("kernel,counter,dispatches,mean,sum")
for k, d in acc.items():
    for c, v in d.items():
        print('"%s",%s,%d,%.6g,%.6g' % (k, c, len(v), sum(v) / len(v), sum(v)))
