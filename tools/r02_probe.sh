#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/probe
mkdir -p $O
for dbg in 0 1 2; do
  BHIP_PROBE_DEBUG=$dbg python bench.py --query q3 --steps 6 --warmup 2 --no-cpu-baseline > $O/q3_dbg${dbg}.json 2> $O/q3_dbg${dbg}.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/q3_dbg${dbg}.json").read())
    print("debug=$dbg ms_per_step=%.3f probe_ms=%.4f" % (d["ms_per_step"], d["roofline"]["kernel_ms"]), d["kernels_ms_per_step"])
except Exception as e:
    print("debug=$dbg failed", e); print(open("$O/q3_dbg${dbg}.err").read()[-800:])
PY
done
