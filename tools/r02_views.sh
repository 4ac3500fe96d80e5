#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/views
mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log | cut -c1-250
