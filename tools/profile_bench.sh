#!/bin/bash
# rocprofv3 evidence for the bench: per-kernel time (--stats) and HBM traffic (PMC, separate passes).
# Run on the GPU box from the repo root; summaries land in gpurun_out/prof_*.
R=$PWD; export TMPDIR=/tmp; cd /tmp
ARGS="--steps ${STEPS:-5} --warmup 2 --no-cpu-baseline ${EXTRA_ARGS}"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_write.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
def agg(tag, ctr):
    f = glob.glob(f"gpurun_out/{tag}/*/*counter_collection.csv")[0]
    rows = list(csv.DictReader(open(f)))
    a = collections.defaultdict(float); d = collections.defaultdict(set)
    for r in rows:
        if r["Counter_Name"] == ctr:
            a[r["Kernel_Name"]] += float(r["Counter_Value"]); d[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return {k: (v / len(d[k]), len(d[k])) for k, v in a.items()}
fs, ws = agg("prof_fetch", "FETCH_SIZE"), agg("prof_write", "WRITE_SIZE")
print("kernel | dispatches | FETCH_SIZE KB/launch (x2 on gfx950 for wide streams) | WRITE_SIZE KB/launch")
for k, (v, n) in sorted(fs.items(), key=lambda kv: -kv[1][0])[:6]:
    print(f"{k[:70]} | {n} | {v:.0f} | {ws.get(k, (0, 0))[0]:.0f}")
import json
top = max(fs.items(), key=lambda kv: kv[1][0])
json.dump({"kernel": top[0], "dispatches": top[1][1], "fetch_size_kb_per_launch": top[1][0],
           "write_size_kb_per_launch": ws.get(top[0], (0, 0))[0],
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; KB = 1024 B; gfx950: FETCH_SIZE x2 for wide streaming reads",
           "query": __import__("os").environ.get("QUERY", "q1"), "rows_per_launch": int(__import__("os").environ.get("ROWS", "600037902"))}, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
st = glob.glob("gpurun_out/prof_stats/*/*kernel_stats.csv")
if st:
    print(open(st[0]).read()[:3000])
PY
