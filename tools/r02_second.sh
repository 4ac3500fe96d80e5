#!/bin/bash
# round 2: fused filter+probe, join column pruning, sample-first aggregate ladder
set -x
export TMPDIR=/tmp
O=gpurun_out/r02_second
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python bench.py --steps 10 --no-cpu-baseline > $O/bench_q1.json 2> $O/bench_q1.err && cat $O/bench_q1.json &&
python bench.py --query q3 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_q3.json 2> $O/bench_q3.err && cat $O/bench_q3.json &&
BHIP_NO_FUSED_PROBE=1 python bench.py --query q3 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_q3_nofused.json 2> $O/bench_q3_nofused.err && cat $O/bench_q3_nofused.json &&
python bench.py --query q5 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_q5.json 2> $O/bench_q5.err && cat $O/bench_q5.json &&
BHIP_KERNEL_TIMING=0 python bench.py --query q3 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_q3_notiming.json 2> $O/bench_q3_notiming.err && cat $O/bench_q3_notiming.json
R=$PWD; cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_q3 -- python3 $R/bench.py --query q3 --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/prof_q3.log 2>&1
cd $R && find $O/prof_q3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/q3_kernel_stats.csv; head -45 $O/q3_kernel_stats.csv
