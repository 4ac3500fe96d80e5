#!/bin/bash
# sweep the fused scan+aggregate kernel variants of a -DBHIP_TUNE build (run on the GPU box)
ROWS=${ROWS:-120000000}
mkdir -p gpurun_out
for r in 1 2 4; do for pf in 0 1; do
  BHIP_SCAN_R=$r BHIP_PREFETCH=$pf timeout -k 10 200 python bench.py --rows $ROWS --steps 8 --warmup 2 --no-cpu-baseline --query ${QUERY:-q1} > gpurun_out/tune.json 2> gpurun_out/tune.err || { echo "R=$r PF=$pf FAILED"; tail -3 gpurun_out/tune.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/tune.json"))
print("R=$r PF=$pf  step %.3f ms  kernel %.3f ms  %.0f GB/s (%.1f%%)  groups %s rows %s" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["achieved"], 100*d["roofline"]["frac"], d["result_check"]["groups"], d["result_check"]["rows_counted"]))
PY
done; done
