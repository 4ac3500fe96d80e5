#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/r02_fourth
mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log | cut -c1-250
python bench.py --steps 10 --no-cpu-baseline > $O/bench_q1.json 2> $O/bench_q1.err && cat $O/bench_q1.json
python bench.py --query q3 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_q3.json 2> $O/bench_q3.err && cat $O/bench_q3.json
python bench.py --query q5 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_q5.json 2> $O/bench_q5.err && cat $O/bench_q5.json
R=$PWD; cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_q1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/prof_q1.log 2>&1
cd $R && find $O/prof_q1 -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/q1_kernel_trace.csv; rm -rf $O/prof_q1
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_q5 -- python3 $R/bench.py --query q5 --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/prof_q5.log 2>&1
cd $R && find $O/prof_q5 -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/q5_kernel_trace.csv; rm -rf $O/prof_q5
