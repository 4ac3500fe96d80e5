#!/bin/bash
R=$PWD; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/prof_tbl -- python3 $R/tools/exp_tbl.py > $R/gpurun_out/prof_tbl.log 2>&1
cd $R
python3 - <<'PY'
import glob, csv
for pat in ("*kernel_stats.csv", "*memory_copy_stats.csv"):
    for f in sorted(glob.glob("gpurun_out/prof_tbl/*/" + pat))[-1:]:
        for r in list(csv.DictReader(open(f)))[:8]:
            print("%-70s calls %4s total %9.2f ms avg %8.3f ms" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
