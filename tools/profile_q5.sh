#!/bin/bash
R=$PWD; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_q5 -- python3 $R/tools/exp_q5.py > $R/gpurun_out/prof_q5.log 2>&1
cd $R
python3 - <<'PY'
import glob, csv
f = sorted(glob.glob("gpurun_out/prof_q5/*/*kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-64s calls %4s total %9.2f ms avg %8.3f ms" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
