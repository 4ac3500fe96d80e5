#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/pair
mkdir -p $O
python -m pytest tests/test_join_paths_gpu.py tests/test_goldens.py -m gpu -q -x > $O/pytest_a.log 2>&1; tail -8 $O/pytest_a.log | cut -c1-250
python bench.py --query q5 --steps 8 --warmup 2 --no-cpu-baseline > $O/q5.json 2> $O/q5.err || { tail -20 $O/q5.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$O/q5.json").read())
print("q5 ms_per_step=%.3f" % d["ms_per_step"], d["kernels_ms_per_step"], d["result_check"])
PY
