#!/bin/bash
# round 2, first GPU pass: tests, bench lines for q1/q3/q5 with fresh plans per step, Q3 kernel stats
set -x
export TMPDIR=/tmp
O=gpurun_out/r02_first
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python bench.py > $O/bench_q1.json 2> $O/bench_q1.err && cat $O/bench_q1.json &&
python bench.py --query q3 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_q3.json 2> $O/bench_q3.err && cat $O/bench_q3.json &&
python bench.py --query q5 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_q5.json 2> $O/bench_q5.err && cat $O/bench_q5.json &&
python bench.py --gpus 2 --backend gloo --rows 100000000 --steps 3 --warmup 1 --no-weak > $O/bench_q1_gloo2.json 2> $O/bench_q1_gloo2.err; cat $O/bench_q1_gloo2.json; tail -5 $O/bench_q1_gloo2.err
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_q3 -o q3 -- python3 $GRAFT_REPO_ROOT/bench.py --query q3 --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_q3.log 2>&1
cd $GRAFT_REPO_ROOT && find $O/prof_q3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/q3_kernel_stats.csv; head -40 $O/q3_kernel_stats.csv
