#!/bin/bash
for a in 0 1 2 3; do
  L=$PWD/ballista_amd/lib/libballista_hip_abl$a.so; [ $a = 0 ] && L=$PWD/ballista_amd/lib/libballista_hip.so
  echo "== ablate $a (1: no string loads in loop, 2: no LDS key match)"
  BHIP_LIB_PATH=$L BHIP_AGG_BLOCKS_PER_CU=3 python tools/exp_shapes.py 2>&1 | grep -E "keys_count_only|q1 "
done
