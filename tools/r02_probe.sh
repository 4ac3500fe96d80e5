#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/probe
mkdir -p $O
python -m pytest tests/test_parquet_gpu.py tests/test_q1_q6_gpu.py tests/test_operators_gpu.py -q -x > $O/pytest_a.log 2>&1; tail -5 $O/pytest_a.log | cut -c1-250
BHIP_PROBE_ROWS=8 python -m pytest tests/test_join_paths_gpu.py tests/test_goldens.py -q -x > $O/pytest_rows8.log 2>&1; tail -5 $O/pytest_rows8.log | cut -c1-250
python bench.py > $O/bench_q1.json 2> $O/bench_q1.err || { tail -20 $O/bench_q1.err; exit 1; }
cut -c1-260 $O/bench_q1.json
for rows in 4 8; do for pc in 4 8 12; do
  BHIP_PROBE_ROWS=$rows BHIP_PROBE_BLOCKS_PER_CU=$pc python bench.py --query q3 --steps 8 --warmup 2 --no-cpu-baseline > $O/q3_r${rows}_pc${pc}.json 2> $O/q3_r${rows}_pc${pc}.err || { tail -20 $O/q3_r${rows}_pc${pc}.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/q3_r${rows}_pc${pc}.json").read())
print("rows=$rows per_cu=$pc ms_per_step=%.3f probe_ms=%.4f frac=%.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))
PY
done; done
