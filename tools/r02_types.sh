#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/types
mkdir -p $O
python -m pytest tests/test_types_gpu.py tests/test_strings_gpu.py -q > $O/pytest_types.log 2>&1; tail -60 $O/pytest_types.log | cut -c1-250
python -m pytest tests -m gpu -q --deselect tests/test_types_gpu.py --deselect tests/test_strings_gpu.py > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log | cut -c1-250
for q in q1 q3 q5; do
python bench.py --query $q --steps 10 --warmup 3 > $O/bench_$q.json 2> $O/bench_$q.err || { tail -20 $O/bench_$q.err; exit 1; }
cut -c1-250 $O/bench_$q.json
done
BHIP_TRACE_HOST=1 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/trace_q1.json 2> $O/trace_q1.err
grep bhip-host $O/trace_q1.err | tail -16
