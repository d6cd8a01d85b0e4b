#!/bin/bash
# one rig frame at a time under rocprofv3 (kernel + HIP API trace): where do the microseconds between the host's graph launch and
# its wake-up go?  Per job: launch call -> first kernel start, kernel durations and the gaps between them, last kernel end -> wake-up.
#   bash scripts/lat_trace.sh <outdir> [VAR=VALUE ...]
O=gpurun_out/${1:-lattrace}; mkdir -p $O; shift
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp && cd $ROOT
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $O/p -o t -- python3 scripts/latency.py --frames 60 > $O/log 2>&1
python3 - <<PY
import csv,glob,collections,statistics as st
kf=glob.glob('$O/p/**/*kernel_trace.csv',recursive=True)[0]
af=glob.glob('$O/p/**/*hip_api_trace.csv',recursive=True)[0]
K=[]
for r in csv.DictReader(open(kf)):
    n=r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ","").replace("mcorb::","")
    K.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),n))
K.sort()
A=[]
for r in csv.DictReader(open(af)):
    A.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Function"]))
A.sort()
L=[a for a in A if a[2]=="hipGraphLaunch"]
if not L:
    L=[a for a in A if a[2] in ("hipLaunchKernel","hipExtLaunchKernel","hipModuleLaunchKernel")]
    print("no hipGraphLaunch in the trace: launch-by-launch run, %d launches" % len(L))
print("graph launches", len(L), "kernels", len(K))
rows=[]
seqs=collections.defaultdict(list)
for i,(s,e,_) in enumerate(L[10:-1], 10):
    nxt=L[i+1][0]
    ks=[k for k in K if k[0]>=s and k[0]<nxt]
    if not ks: continue
    sync=[a for a in A if a[0]>=e and a[0]<nxt and a[2] in ("hipEventSynchronize","hipEventQuery","hipStreamSynchronize")]
    wake=max((a[1] for a in sync if a[2]=="hipEventSynchronize"), default=None)
    busy=sum(k[1]-k[0] for k in ks)
    gaps=sum(max(0,ks[j+1][0]-ks[j][1]) for j in range(len(ks)-1))
    rows.append((e-s, ks[0][0]-s, ks[-1][1]-ks[0][0], busy, gaps, (wake-ks[-1][1]) if wake else 0, len(ks)))
    for j,k in enumerate(ks): seqs[j].append((k[2],k[1]-k[0],(ks[j][0]-ks[j-1][1]) if j else 0))
med=lambda c:[st.median(r[c] for r in rows)/1e3][0]
print("per job (median of %d): launch call %.1f us | call start -> first kernel %.1f | first kernel start -> last kernel end %.1f (kernels %.1f + gaps %.1f, %d kernels) | last kernel end -> hipEventSynchronize returns %.1f" % (len(rows),med(0),med(1),med(2),med(3),med(4),rows[0][6],med(5)))
for j in sorted(seqs):
    n=collections.Counter(x[0] for x in seqs[j]).most_common(1)[0][0]
    print("  %2d %-22s %6.1f us  gap before %5.1f" % (j,n,st.median(x[1] for x in seqs[j])/1e3,st.median(x[2] for x in seqs[j])/1e3))
PY
tail -1 $O/log
rm -rf $O/p
