#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/r02_ninth
mkdir -p $O
python -m pytest tests/test_parquet_gpu.py tests/test_join_paths_gpu.py tests/test_goldens.py -m gpu -q > $O/pytest_new.log 2>&1; tail -8 $O/pytest_new.log | cut -c1-250
BHIP_JOIN_RADIX=1 python -m pytest tests/test_join_paths_gpu.py tests/test_goldens.py -m gpu -q > $O/pytest_radix.log 2>&1; tail -8 $O/pytest_radix.log | cut -c1-250
for mode in rank radix table; do
  case $mode in rank) E="";; radix) E="BHIP_JOIN_RADIX=1";; table) E="BHIP_JOIN_TABLE=1";; esac
  env $E python bench.py --query q3 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_q3_$mode.json 2> $O/bench_q3_$mode.err; python -c "
import json;d=json.load(open('$O/bench_q3_$mode.json'));print('q3 $mode', round(d['ms_per_step'],3),'ms', d['kernels_ms_per_step'])"; done
R=$PWD; cd /tmp && BHIP_JOIN_RADIX=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_radix -- python3 $R/bench.py --query q3 --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/prof_radix.log 2>&1
cd $R && find $O/prof_radix -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/q3_radix_kernel_stats.csv; rm -rf $O/prof_radix; head -12 $O/q3_radix_kernel_stats.csv | cut -c1-200
