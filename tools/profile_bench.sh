#!/bin/bash
# rocprofv3 evidence for the bench: per-kernel time (--stats) and HBM traffic (PMC, separate passes, counters only with --kernel-trace).
# Run on the GPU box from the repo root:  QUERY=q1|q6|q3|q5 bash tools/profile_bench.sh ; summaries land in gpurun_out/prof_<query>*.
# Copy what should be judged into profiles/ (tracked).
R=$PWD; export TMPDIR=/tmp; cd /tmp
Q=${QUERY:-q1}
ARGS="--query $Q --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline ${EXTRA_ARGS}"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${Q}_stats -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${Q}_stats.log 2>&1
if [ -z "$STATS_ONLY" ]; then          # STATS_ONLY=1: the kernel table and the step timeline, no counter passes
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${Q}_fetch -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${Q}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${Q}_write -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${Q}_write.log 2>&1
else rm -rf $R/gpurun_out/prof_${Q}_fetch $R/gpurun_out/prof_${Q}_write; fi
cd $R
QUERY=$Q python3 - <<'PY'
import csv, glob, collections, json, os, shutil
q = os.environ["QUERY"]
def agg(tag, ctr):
    fl = glob.glob(f"gpurun_out/{tag}/*/*counter_collection.csv")
    if not fl: return {}
    f = fl[0]
    a = collections.defaultdict(float); d = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr:
            a[r["Kernel_Name"]] += float(r["Counter_Value"]); d[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return {k: (v / len(d[k]), len(d[k])) for k, v in a.items()}
fs, ws = agg(f"prof_{q}_fetch", "FETCH_SIZE"), agg(f"prof_{q}_write", "WRITE_SIZE")
st = glob.glob(f"gpurun_out/prof_{q}_stats/*/*kernel_stats.csv")[0]
shutil.copy(st, f"gpurun_out/{q}_kernel_stats.csv")
stats = [r for r in csv.DictReader(open(st)) if "gen_" not in r["Name"]]
stats.sort(key=lambda r: -float(r["TotalDurationNs"]))
lines = ["kernel | calls | avg ms | FETCH_SIZE KB/launch (x2 on gfx950 for wide streams) | WRITE_SIZE KB/launch"]
for r in stats[:12]:
    k = r["Name"]
    lines.append(f"{k[:90]} | {r['Calls']} | {float(r['AverageNs']) / 1e6:.4f} | {fs.get(k, (0, 0))[0]:.0f} | {ws.get(k, (0, 0))[0]:.0f}")
open(f"gpurun_out/{q}_pmc_traffic.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
# the launches of the last timed step, in stream order: start offset and duration (us) — where a step's time outside its big kernel goes
tr = glob.glob(f"gpurun_out/prof_{q}_stats/*/*kernel_trace.csv")
if tr:
    rows = sorted(csv.DictReader(open(tr[0])), key=lambda r: int(r["Start_Timestamp"]))
    big = [i for i, r in enumerate(rows) if r["Kernel_Name"] == stats[0]["Name"]]
    per_step = max(1, len(big) // max(1, int(os.environ.get("STEPS_TOTAL", "0")) or len(big)))
    if len(big) >= 2:
        a = big[-1 - per_step] if len(big) > per_step else big[0]
        t0 = int(rows[a]["Start_Timestamp"])
        tl = ["start_us | dur_us | kernel   (from the first dominant-kernel launch of the second-to-last step to the end of the run)"]
        for r in rows[a:]:
            tl.append(f"{(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} | {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} | {r['Kernel_Name'][:100]}")
        open(f"gpurun_out/{q}_step_timeline.txt", "w").write("\n".join(tl) + "\n")
top = stats[0]["Name"]
# the template instantiations of the dominant kernel count as ONE kernel (bench.py times them under one name): launch-weighted averages
import re
base = re.split(r"[<(]", top.replace("(anonymous namespace)::", ""))[0].split("::")[-1].split()[-1]
same = [r for r in stats if base in r["Name"]]
calls = sum(int(r["Calls"]) for r in same)
def wavg(m):
    tot = sum(m.get(r["Name"], (0, 0))[0] * m.get(r["Name"], (0, 0))[1] for r in same)
    n = sum(m.get(r["Name"], (0, 0))[1] for r in same)
    return tot / n if n else 0.0
if fs: json.dump({"kernel": base, "variants": [r["Name"][:120] for r in same], "dispatches": calls,
           "avg_ms": sum(float(r["TotalDurationNs"]) for r in same) / max(calls, 1) / 1e6,
           "fetch_size_kb_per_launch": wavg(fs), "write_size_kb_per_launch": wavg(ws),
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; KB = 1024 B; gfx950: FETCH_SIZE x2 for wide streaming reads; "
                   "average over the kernel's launches of the run (a join query launches it once per probe side)",
           "query": q, "rows_per_launch": int(os.environ.get("ROWS", "600037902"))}, open(f"gpurun_out/pmc_traffic_{q}.json", "w"), indent=1)
PY
