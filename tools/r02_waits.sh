#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/waits
mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log | cut -c1-250
for q in q1 q3 q5; do
  python bench.py --query $q --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$q.json 2> $O/bench_$q.err || { tail -20 $O/bench_$q.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/bench_$q.json").read())
print("$q ms_per_step=%.3f kernel=%s %.4f frac=%.3f" % (d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))
PY
done
