#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/views
mkdir -p $O
python -m pytest tests/test_join_paths_gpu.py tests/test_goldens.py tests/test_operators_gpu.py tests/test_q1_q6_gpu.py tests/test_full_size_gpu.py -m gpu -q -x > $O/pytest_a.log 2>&1; tail -6 $O/pytest_a.log | cut -c1-250
for q in q1 q3 q5; do
  python bench.py --query $q --steps 8 --warmup 2 --no-cpu-baseline > $O/${q}.json 2> $O/${q}.err || { tail -20 $O/${q}.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/${q}.json").read())
print("$q ms_per_step=%.3f" % d["ms_per_step"], d["kernels_ms_per_step"])
PY
done
