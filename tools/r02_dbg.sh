#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02_dbg
mkdir -p $O
python -m pytest tests/test_join_paths_gpu.py tests/test_goldens.py -m gpu -q > $O/pytest_rank.log 2>&1; tail -40 $O/pytest_rank.log | cut -c1-200
echo "=== with BHIP_JOIN_TABLE=1"
BHIP_JOIN_TABLE=1 python -m pytest tests/test_join_paths_gpu.py tests/test_goldens.py -m gpu -q > $O/pytest_table.log 2>&1; tail -15 $O/pytest_table.log | cut -c1-200
