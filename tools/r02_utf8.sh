#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/utf8
mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log | cut -c1-250
python bench.py --query q5 --steps 8 --warmup 2 --no-cpu-baseline > $O/q5.json 2> $O/q5.err || { tail -20 $O/q5.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$O/q5.json").read())
print("q5 ms_per_step=%.3f" % d["ms_per_step"], d["kernels_ms_per_step"])
PY
