#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/r02_third
mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python bench.py --query q3 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_q3.json 2> $O/bench_q3.err && cat $O/bench_q3.json &&
BHIP_JOIN_TABLE=1 python bench.py --query q3 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_q3_table.json 2> $O/bench_q3_table.err && cat $O/bench_q3_table.json &&
python bench.py --query q5 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_q5.json 2> $O/bench_q5.err && cat $O/bench_q5.json &&
BHIP_PROBE_BLOCKS_PER_CU=16 python bench.py --query q3 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_q3_b16.json 2> $O/bench_q3_b16.err && cat $O/bench_q3_b16.json
R=$PWD; cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_q3 -- python3 $R/bench.py --query q3 --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/prof_q3.log 2>&1
cd $R && find $O/prof_q3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/q3_kernel_stats.csv; find $O/prof_q3 -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/q3_kernel_trace.csv; rm -rf $O/prof_q3; head -30 $O/q3_kernel_stats.csv | cut -c1-180
