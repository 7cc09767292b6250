#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
TPA_SVD_DEBUG_ACT=1 timeout 900 python bench.py --config hubbard1024 --steps 1 --warmup 5 --no-cpu-baseline --no-extras > $O/act_hub.log 2> $O/act_hub.err
grep svd_act $O/act_hub.err | tail -2500 > $O/act_hub_tail.txt
python - <<'P'
import re,collections
lines=open('gpurun_out/r06/act_hub_tail.txt').read().splitlines()
# group into calls: a call starts at "sweep 0"
calls=[];cur=[]
for l in lines:
    m=re.match(r'svd_act sweep (\d+) jobs (\d+):(.*)',l)
    if not m: continue
    sw=int(m.group(1))
    if sw==0 and cur: calls.append(cur); cur=[]
    cur.append((sw,int(m.group(2)),m.group(3).split()))
if cur: calls.append(cur)
print(len(calls),'calls; sweeps per call histogram:',collections.Counter(len(c) for c in calls))
# for the last sweeps of each call: which R sizes are still active
late=collections.Counter(); lateb=collections.Counter()
for c in calls:
    for sw,nj,items in c[3:]:
        for it in items:
            R=int(it[1:].split(':')[0]); late[min(R//64*64,512)]+=1
            if it.endswith('b'): lateb[min(R//64*64,512)]+=1
print('jobs active in sweeps >= 3, by R bucket (64-row buckets):',sorted(late.items()))
print('... with big pairs:',sorted(lateb.items()))
for c in calls[-3:]:
    print('--- call with',c[0][1],'jobs')
    for sw,nj,items in c: print('  sweep',sw,' '.join(items[:40]))
P
