#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02n
mkdir -p $OUT
cd $R
timeout 2000 python tools/tune_sweep.py --repeat 2 base conv_t3=RFE_CONV_TALL=3 conv_t4=RFE_CONV_TALL=4 gemm_sl2=RFE_GEMM_SLEEP=2 gemm_sl8=RFE_GEMM_SLEEP=8 att_sl4=RFE_ATT_PRIO=5 att_sl16=RFE_ATT_PRIO=17 2>&1 | tee $OUT/sweep.txt | cut -c1-330
