#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02o
mkdir -p $OUT
cd $R
RFE_LIBRARY=$R/rover-slam_amd/librover_fe_tuning.so RFE_GEMM_DB=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -4
timeout 1500 python tools/tune_sweep.py --repeat 3 base gemm_db=RFE_GEMM_DB=1 2>&1 | tee $OUT/sweep.txt | cut -c1-330
