#!/bin/bash
# One GPU-box call that re-establishes the round's evidence: the GPU parity suite, the four bench lines, and the rocprofv3
# kernel-stats + PMC summaries of the queries named in PROFILE (default "q1 q3").  Results: gpurun_out/round_check/.
# Run from the repo root:  bash tools/round_check.sh          (SKIP_TESTS=1 leaves the suite out)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/round_check
mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then
  python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
  tail -3 $O/pytest_gpu.log
fi
python bench.py > $O/bench_q1.json 2> $O/bench_q1.err || { tail -20 $O/bench_q1.err; exit 1; }
for q in q3 q5 q6; do
  python bench.py --query $q --steps 8 --warmup 2 > $O/bench_$q.json 2> $O/bench_$q.err || { tail -20 $O/bench_$q.err; exit 1; }
done
cut -c1-1500 $O/bench_q*.json
for q in ${PROFILE:-q1 q3}; do
  QUERY=$q STEPS=${STEPS:-4} bash tools/profile_bench.sh > $O/profile_$q.log 2>&1 || { tail -20 $O/profile_$q.log; exit 1; }
  tail -14 $O/profile_$q.log | cut -c1-200
  cp gpurun_out/pmc_traffic_$q.json gpurun_out/${q}_kernel_stats.csv gpurun_out/${q}_pmc_traffic.txt $O/
  rm -rf gpurun_out/prof_${q}_stats gpurun_out/prof_${q}_fetch gpurun_out/prof_${q}_write
done
