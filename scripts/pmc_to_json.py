"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (counter_collection CSVs) of scripts/svd_file_bench.py -> the JSON that
bench.py quotes as roofline.traffic:  python scripts/pmc_to_json.py <fetch.csv> <write.csv> <calls> <out.json>
FETCH_SIZE is doubled (gfx950: 128-B requests are tallied at 64 B for wide coalesced reads, MI355X_MICROARCH.md, HBM section);
WRITE_SIZE is taken as is (uncalibrated there).  Both are fabric-side (TCC_EA) counters in KB: Infinity-Cache hits are included."""
import csv
import json
import sys
from collections import defaultdict

fetch_csv, write_csv, calls, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]


def read(path, counter):
    acc, n = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row['Counter_Name'] != counter:
                continue
            k = row['Kernel_Name']
            k = k.replace('(anonymous namespace)::', '').replace('void ', '')
            k = k.split('(')[0].split('<')[0].split('::')[-1] if '::' in k else k.split('(')[0]
            acc[k] += float(row['Counter_Value'])
            n[k] += 1
    return acc, n


fa, fn = read(fetch_csv, 'FETCH_SIZE')
wa, wn = read(write_csv, 'WRITE_SIZE')
per = {}
for k in sorted(set(fa) | set(wa)):
    per[k] = {"FETCH_SIZE_KB_per_call": fa.get(k, 0.) / calls, "WRITE_SIZE_KB_per_call": wa.get(k, 0.) / calls,
              "dispatches_per_call": max(fn.get(k, 0), wn.get(k, 0)) / calls}
F, W = sum(fa.values()) / calls, sum(wa.values()) / calls
res = {"what": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of scripts/svd_file_bench.py: %d calls of tpa_svd_batch on the "
               "saturated chi=2048 centre-bond theta of the heis2048 workload (10 charge blocks, largest ~1080 x 1080, f64), cold path "
               "(pivoted QR + Gram-only sweeps on 32-row blocks, one launch per round; round 6: floor 1e-2 on the smaller row, "
               "activity-driven rounds)" % calls,
       "calls": calls, "FETCH_SIZE_KB_per_call_raw": F, "WRITE_SIZE_KB_per_call_raw": W,
       "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM "
                     "section); WRITE_SIZE uncalibrated, taken as is",
       "bytes_per_call_corrected": (2. * F + W) * 1024.,
       "note": "fabric-side (TCC_EA) counters: Infinity-Cache hits are included, so this is mostly on-die re-streaming of the row "
               "blocks between the launches of a Jacobi round, not HBM traffic",
       "per_kernel": per}
if len(sys.argv) > 5:      # the dumped theta: algorithmic bytes / flops of the SAME call (SURVEY 8(d) model per charge block)
    import numpy as np
    d = np.load(sys.argv[5])
    nb, fl = 0., 0.
    for k in d.files:
        m, n = d[k].shape
        big, small = float(max(m, n)), float(min(m, n))
        nb += 8. * (big * small * 3. + small)
        fl += 4. * big * big * small + 8. * big * small * small + 9. * small ** 3
    res["algorithmic_bytes_same_call"] = nb
    res["algorithmic_flops_same_call"] = fl
    res["traffic_over_algorithmic"] = res["bytes_per_call_corrected"] / nb
with open(out, 'w') as f:
    json.dump(res, f, indent=1)
print(json.dumps({k: res[k] for k in ('calls', 'FETCH_SIZE_KB_per_call_raw', 'WRITE_SIZE_KB_per_call_raw', 'bytes_per_call_corrected')}))
