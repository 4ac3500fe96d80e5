#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/r02_tenth
mkdir -p $O
python bench.py > $O/bench_q1_default.json 2> $O/bench_q1_default.err; cat $O/bench_q1_default.json | cut -c1-3000
python bench.py --query q3 --steps 8 --warmup 2 > $O/bench_q3_full.json 2> $O/bench_q3_full.err; cat $O/bench_q3_full.json | cut -c1-3000
python bench.py --query q5 --steps 8 --warmup 2 > $O/bench_q5_full.json 2> $O/bench_q5_full.err; cat $O/bench_q5_full.json | cut -c1-3000
python bench.py --query q6 --steps 10 --warmup 2 > $O/bench_q6_full.json 2> $O/bench_q6_full.err; cat $O/bench_q6_full.json | cut -c1-2000
QUERY=q3 STEPS=3 bash tools/profile_bench.sh > $O/profile_q3.log 2>&1; tail -15 $O/profile_q3.log | cut -c1-220
QUERY=q1 STEPS=5 bash tools/profile_bench.sh > $O/profile_q1.log 2>&1; tail -15 $O/profile_q1.log | cut -c1-220
cp gpurun_out/pmc_traffic_q3.json gpurun_out/pmc_traffic_q1.json gpurun_out/q3_kernel_stats.csv gpurun_out/q1_kernel_stats.csv gpurun_out/q3_pmc_traffic.txt gpurun_out/q1_pmc_traffic.txt $O/ 2>/dev/null
rm -rf gpurun_out/prof_q*_stats gpurun_out/prof_q*_fetch gpurun_out/prof_q*_write
