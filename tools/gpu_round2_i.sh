#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02i
mkdir -p $OUT
cd $R
timeout 1500 python tools/tune_sweep.py --repeat 2 base gelu_fast=RFE_GELU_FAST=1 2>&1 | tee $OUT/sweep.txt | cut -c1-330
RFE_LIBRARY=$R/rover-slam_amd/librover_fe_tuning.so RFE_GELU_FAST=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "lightglue or stream or full_size" 2>&1 | tail -5
RFE_LIBRARY=$R/rover-slam_amd/librover_fe_tuning.so RFE_GELU_FAST=1 timeout 1500 python tools/lg_tolerance_study.py --cases 20 > $OUT/lg_tol_gelu_fast.md 2> /dev/null; tail -2 $OUT/lg_tol_gelu_fast.md
