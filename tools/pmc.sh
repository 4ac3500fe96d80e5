#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" <bench args...>   (run on the GPU box; output under gpurun_out/<tag>)
TAG=$1; CTRS=$2; shift 2
R=$PWD; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $R/gpurun_out/$TAG -- python3 $R/${PMC_SCRIPT:-bench.py} "$@" > $R/gpurun_out/$TAG.log 2>&1
cd $R
python3 - <<PY
import csv, collections, glob
f=glob.glob("gpurun_out/$TAG/*/*counter_collection.csv")[0]
rows=list(csv.DictReader(open(f)))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); disp=collections.defaultdict(set)
for r in rows:
    k=r['Kernel_Name'][:int('${PMC_NAME_CHARS:-48}')]; agg[k][r['Counter_Name']]+=float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
kt=list(csv.DictReader(open(glob.glob("gpurun_out/$TAG/*/*kernel_trace.csv")[0])))
d=collections.defaultdict(list)
for r in kt: d[r['Kernel_Name'][:int('${PMC_NAME_CHARS:-48}')]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
for k,v in agg.items():
    n=len(disp[k])
    if sum(d[k])/len(d[k]) < 0.05: continue
    print(k, 'n=%d avg_ms=%.3f'%(n, sum(d[k])/len(d[k])), {c: round(x/max(n,1)) for c,x in v.items()})
PY
