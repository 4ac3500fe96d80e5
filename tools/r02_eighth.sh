#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/r02_eighth
mkdir -p $O
python -m pytest tests/test_parquet_gpu.py tests/test_ipc.py tests/test_exchange_gpu.py -m gpu -q > $O/pytest_new.log 2>&1; tail -30 $O/pytest_new.log | cut -c1-250
for q in q1 q3 q5; do python bench.py --query $q --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_$q.json 2> $O/bench_$q.err; python -c "
import json;d=json.load(open('$O/bench_$q.json'));print('$q', round(d['ms_per_step'],3),'ms', d['roofline'])"; done
