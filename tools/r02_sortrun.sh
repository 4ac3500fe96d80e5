#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/sortrun
mkdir -p $O
python -m pytest tests/test_operators_gpu.py tests/test_goldens.py -q -x > $O/pytest_a.log 2>&1; tail -6 $O/pytest_a.log | cut -c1-250
for f in 0 1; do
  BHIP_NO_FUSED_RADIX=$f python bench.py --query q3 --steps 8 --warmup 2 --no-cpu-baseline > $O/q3_nofused$f.json 2> $O/q3_nofused$f.err || { tail -20 $O/q3_nofused$f.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/q3_nofused$f.json").read())
print("BHIP_NO_FUSED_RADIX=$f q3 ms_per_step=%.3f" % d["ms_per_step"])
PY
done
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sortrun/prof -- python3 $GRAFT_REPO_ROOT/bench.py --query q3 --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/sortrun/prof.log 2>&1
cd $GRAFT_REPO_ROOT && python - <<'PY'
import csv, glob
f=glob.glob("gpurun_out/sortrun/prof/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "radix" in r["Name"] or "scan_" in r["Name"]:
        print(r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3)
PY
rm -rf gpurun_out/sortrun/prof
