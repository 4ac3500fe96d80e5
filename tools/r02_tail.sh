#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/tail
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
python bench.py > $O/bench_q1.json 2> $O/bench_q1.err || { tail -20 $O/bench_q1.err; exit 1; }
cut -c1-300 $O/bench_q1.json
python bench.py --query q5 --steps 8 --warmup 2 > $O/bench_q5.json 2> $O/bench_q5.err || { tail -20 $O/bench_q5.err; exit 1; }
cut -c1-300 $O/bench_q5.json
QUERY=q1 STEPS=4 bash tools/profile_bench.sh > $O/profile_q1.log 2>&1 || { tail -20 $O/profile_q1.log; exit 1; }
cp gpurun_out/q1_step_timeline.txt gpurun_out/q1_kernel_stats.csv $O/
tail -40 gpurun_out/q1_step_timeline.txt | cut -c1-150
rm -rf gpurun_out/prof_q1_stats gpurun_out/prof_q1_fetch gpurun_out/prof_q1_write
