#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/runs
mkdir -p $O
python -m pytest tests/test_operators_gpu.py tests/test_goldens.py tests/test_full_size_gpu.py tests/test_q12_gpu.py tests/test_q1_q6_gpu.py -q -x > $O/pytest_a.log 2>&1; tail -8 $O/pytest_a.log | cut -c1-250
for f in 0 1; do
  BHIP_NO_RUN_AGG=$f python bench.py --query q3 --steps 8 --warmup 2 --no-cpu-baseline > $O/q3_noruns$f.json 2> $O/q3_noruns$f.err || { tail -20 $O/q3_noruns$f.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/q3_noruns$f.json").read())
print("BHIP_NO_RUN_AGG=$f q3 ms_per_step=%.3f" % d["ms_per_step"], d["kernels_ms_per_step"])
PY
done
