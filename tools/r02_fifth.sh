#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/r02_fifth
mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log | cut -c1-250
python bench.py --query q3 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_q3.json 2> $O/bench_q3.err && cat $O/bench_q3.json
BHIP_PROBE_BLOCKS_PER_CU=16 python bench.py --query q3 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_q3_b16.json 2> $O/bench_q3_b16.err && cat $O/bench_q3_b16.json
python bench.py --query q5 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_q5.json 2> $O/bench_q5.err && cat $O/bench_q5.json
