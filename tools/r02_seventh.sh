#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/r02_seventh
mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log | cut -c1-250
for q in q3 q5; do python bench.py --query $q --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_$q.json 2> $O/bench_$q.err; python -c "
import json;d=json.load(open('$O/bench_$q.json'));print('$q', round(d['ms_per_step'],3),'ms', d['roofline'], d['kernels_ms_per_step'])"; done
BHIP_AGG_ATOMIC=1 python bench.py --query q3 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_q3_atomic.json 2> $O/bench_q3_atomic.err; python -c "
import json;d=json.load(open('$O/bench_q3_atomic.json'));print('q3 atomic', round(d['ms_per_step'],3),'ms', d['kernels_ms_per_step'])"
