#!/bin/bash
# One GPU-box call that re-establishes the round's evidence: the GPU parity suite, the default bench line (four queries), and the rocprofv3
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
# the default run: Q1 as metric / value / roofline / cpu_baseline, Q6 / Q3 / Q5 under `configs` (what the driver records)
python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
cut -c1-1500 $O/bench_default.json
for q in ${PROFILE:-q1 q3}; do
  QUERY=$q STEPS=${STEPS:-4} EXTRA_ARGS="--configs=" bash tools/profile_bench.sh > $O/profile_$q.log 2>&1 || { tail -20 $O/profile_$q.log; exit 1; }
  tail -14 $O/profile_$q.log | cut -c1-200
  cp gpurun_out/pmc_traffic_$q.json gpurun_out/${q}_kernel_stats.csv gpurun_out/${q}_pmc_traffic.txt $O/
  rm -rf gpurun_out/prof_${q}_stats gpurun_out/prof_${q}_fetch gpurun_out/prof_${q}_write
done
