#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/r02_sixth
mkdir -p $O
python -m pytest tests -m gpu -q -x --deselect tests/test_full_size_gpu.py::test_q3_is_bit_reproducible_run_to_run > $O/pytest.log 2>&1; tail -5 $O/pytest.log | cut -c1-250
for q in q1 q3 q5; do python bench.py --query $q --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_$q.json 2> $O/bench_$q.err; python -c "
import json;d=json.load(open('$O/bench_$q.json'));print('$q', round(d['ms_per_step'],3),'ms', d['kernels_ms_per_step'])"; done
for q in q1 q3 q5; do BHIP_SPIN_WAIT=0 python bench.py --query $q --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_${q}_nospin.json 2> $O/bench_${q}_nospin.err; python -c "
import json;d=json.load(open('$O/bench_${q}_nospin.json'));print('$q nospin', round(d['ms_per_step'],3),'ms')"; done
