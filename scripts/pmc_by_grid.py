"""Summarise a rocprofv3 --pmc counter_collection.csv per (kernel, grid size): dispatch count and mean counter value per launch.
The grouped GEMM launches of a matvec differ only by their grid (tiles x 256 threads), so the grid size tells step 1 from step 2 and a
split-K launch from an unsplit one.   python scripts/pmc_by_grid.py <counter_collection.csv> [substring of the kernel name]"""
import csv
import sys
from collections import defaultdict
path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ''
acc = defaultdict(lambda: defaultdict(list))
with open(path) as f:
    for row in csv.DictReader(f):
        name = row['Kernel_Name']
        if want not in name:
            continue
        short = name.replace('(anonymous namespace)::', '').split('(')[0][:70]
        acc[(short, int(row.get('Grid_Size', 0) or 0))][row['Counter_Name']].append(float(row['Counter_Value']))
for (k, grid), d in sorted(acc.items()):
    n = max(len(v) for v in d.values())
    print("%s  grid %d (%d workgroups of 256)  dispatches %d" % (k, grid, grid // 256, n))
    wave = d.get('SQ_WAVE_CYCLES')
    for c, v in sorted(d.items()):
        mean = sum(v) / len(v)
        frac = "   %.3f of SQ_WAVE_CYCLES" % (mean / (sum(wave) / len(wave))) if wave and c != 'SQ_WAVE_CYCLES' and c.startswith('SQ_') else ''
        print("   %-28s %.4g%s" % (c, mean, frac))
