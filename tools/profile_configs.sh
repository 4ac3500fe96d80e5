#!/bin/bash
# rocprofv3 kernel stats of tools/exp_configs.py (Q6 + Filter + Q3); run on the GPU box from the repo root
R=$PWD; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg -- python3 $R/tools/exp_configs.py > $R/gpurun_out/prof_cfg.log 2>&1
cd $R
python3 - <<'PY'
import glob
st = sorted(glob.glob("gpurun_out/prof_cfg/*/*kernel_stats.csv"))
print(open(st[-1]).read()[:6000])
PY
