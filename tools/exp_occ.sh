#!/bin/bash
# occupancy / rows-per-thread sensitivity of the SOP kernel (run on the GPU box)
run() { python bench.py --rows 120000000 --steps 8 --warmup 2 --no-cpu-baseline --query $3 > gpurun_out/e.json 2> gpurun_out/e.err && python -c "import json;d=json.load(open('gpurun_out/e.json'));print('$1 blocks/CU=$2 $3 kernel %.3f ms %.0f GB/s' % (d['roofline']['kernel_ms'], d['roofline']['achieved']))" || tail -2 gpurun_out/e.err; }
for q in q1 q6; do
for b in 1 2 3 4; do BHIP_AGG_BLOCKS_PER_CU=$b run R2 $b $q; done
for b in 1 2; do BHIP_LIB_PATH=$PWD/ballista_amd/lib/libballista_hip_r4.so BHIP_AGG_BLOCKS_PER_CU=$b run R4 $b $q; done
done
