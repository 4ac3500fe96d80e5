#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/trace
mkdir -p $O
for q in q3 q5; do
BHIP_TRACE_HOST=1 python bench.py --query $q --steps 2 --warmup 2 --no-cpu-baseline > $O/trace_$q.json 2> $O/trace_$q.err
grep bhip-host $O/trace_$q.err > $O/trace_$q.txt
python - <<PY
lines=[l.split(None,2) for l in open("$O/trace_$q.txt")]
ev=[(float(l[1]), l[2].strip()) for l in lines]
# last step: from the last "plan_collect: enter"
starts=[i for i,e in enumerate(ev) if e[1]=="plan_collect: enter"]
a=starts[-1]
t0=ev[a][0]
waits=0; waited=0.0; last_enter=None
for t,w in ev[a:]:
    if w=="wait: enter": last_enter=t; waits+=1
    if w=="wait: leave" and last_enter is not None: waited+=t-last_enter
print("$q", "step_us=%.0f waits=%d time_in_waits_us=%.0f" % (ev[-1][0]-t0, waits, waited))
prev=t0
for t,w in ev[a:]:
    if w=="wait: leave": print("   wait done at +%.0f (waited %.0f)" % (t-t0, t-prev))
    if w=="wait: enter": prev=t
PY
done
