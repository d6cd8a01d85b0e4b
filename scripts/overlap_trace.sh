#!/bin/bash
# kernel timeline of the default pipeline (6 slots): how much do kernels of different streams overlap in time?
#   bash scripts/overlap_trace.sh <outdir> [VAR=VALUE ...]
O=gpurun_out/${1:-ovl}; mkdir -p $O; shift
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp && cd $ROOT
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/p -o t -- python3 bench.py --no-cpu --no-latency --no-staging --host-cores 0 --no-extra-legs --repeats 1 --iso-jobs 0 --min-region-s 0.3 > $O/log 2>&1
python3 - <<PY
import csv,glob,json,collections
f=glob.glob('$O/p/**/*kernel_trace.csv',recursive=True)[0]
rows=[]
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ","").replace("mcorb::","")
    rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),n,r.get("Queue_Id","?")))
rows.sort()
# steady state: the last 60 % of the trace
t0=rows[int(len(rows)*0.4)][0]; rows=[r for r in rows if r[0]>=t0]
span=rows[-1][1]-rows[0][0]
busy=sum(e-s for s,e,_,_ in rows)
# union of intervals
u=0; cs,ce=rows[0][0],rows[0][1]
for s,e,_,_ in rows[1:]:
    if s>ce: u+=ce-cs; cs,ce=s,e
    else: ce=max(ce,e)
u+=ce-cs
queues=collections.Counter(q for *_,q in rows)
print("kernels %d, queues %s" % (len(rows), dict(queues)))
print("span %.1f ms, sum of kernel durations %.1f ms (%.2f x span), union of kernel intervals %.1f ms (%.1f %% of span busy)" % (span/1e6, busy/1e6, busy/span, u/1e6, 100*u/span))
per=collections.defaultdict(lambda:[0,0])
for s,e,n,_ in rows: per[n][0]+=1; per[n][1]+=e-s
for n,(c,t) in sorted(per.items(),key=lambda x:-x[1][1]): print("  %-24s n %5d avg %7.1f us  share of span %.1f %%"%(n,c,t/c/1e3,100*t/span))
for l in open('$O/log'):
    if l.startswith('{'): print("value", json.loads(l)["value"])
PY
rm -rf $O/p
