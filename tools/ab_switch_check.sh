#!/bin/bash
# Every A/B switch of the library (tools/README.md) must stay parity-green: the affected part of the GPU suite under each of them.
# Run on the GPU box from the repo root; one line per switch in gpurun_out/ab_switch_check.txt.
export TMPDIR=/tmp
O=gpurun_out/ab_switch_check.txt
: > $O
J="tests/test_join_paths_gpu.py tests/test_goldens.py tests/test_operators_gpu.py tests/test_q12_gpu.py"
A="tests/test_operators_gpu.py tests/test_goldens.py tests/test_q1_q6_gpu.py tests/test_lean_gpu.py tests/test_types_gpu.py tests/test_strings_gpu.py"
run() { # name=value, test list
  local out; out=$(env "$1" python -m pytest $2 -m gpu -q 2>&1 | tail -1); echo "$1 : $out" | tee -a $O
}
if [ -z "$NEW_ONLY" ]; then          # NEW_ONLY=1: the switches added in round 3 only
run BHIP_SPIN_WAIT=0 "$J $A"
run BHIP_NO_JOIN_VIEWS=1 "$J"
run BHIP_JOIN_TABLE=1 "$J"
run BHIP_JOIN_RADIX=1 "$J"
run BHIP_NO_NARROW_JOIN=1 "$J"
run BHIP_NO_FUSED_PROBE=1 "$J"
run BHIP_PROBE_ROWS=4 "tests/test_join_paths_gpu.py"
run BHIP_NO_RUN_AGG=1 "$A"
run BHIP_AGG_ATOMIC=1 "$A"
run BHIP_NO_FINAL_ELISION=1 "$A"
run BHIP_NO_EARLY_EMIT=1 "$A"
run BHIP_NO_ROWSORT=1 "$A"
run BHIP_NO_LEAN=1 "$A"
run BHIP_NO_SOP=1 "$A"
run BHIP_NO_RANGE_FILTER=1 "$A"
run BHIP_NO_PARTITION_SCATTER=1 "tests/test_repartition_gpu.py tests/test_operators_gpu.py"
fi
# round 3 (tests that assert WHICH path ran are left out under the switch that turns that path off)
run BHIP_NO_DISTINCT_RUNS=1 "$A"
run BHIP_PROBE_MAP_FLAT=1 "$J"
run BHIP_PROBE_NO_BITS=1 "$J"
run BHIP_PROBE_NO_SCALAR_MAP=1 "$J"
run BHIP_NO_BUCKET_SORT=1 "$A -k not(bucket_path)"
run BHIP_NO_UTF8_EQ_FILTER=1 "$A"
run BHIP_NO_SLOT_EMIT=1 "$A"
run BHIP_NO_FIXED_KEY_PACK=1 "$A"
run BHIP_NO_TINY_BUILD=1 "$J"
run BHIP_NO_STREAMING_SHUFFLE=1 "tests/test_exchange_gpu.py -k not(streaming_shuffle)"
run BHIP_PARQUET_PAGEABLE=1 "tests/test_parquet_gpu.py"
run BHIP_PARQUET_PER_PAGE=1 "tests/test_parquet_gpu.py"
