#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02c
mkdir -p $OUT
cd $R
timeout 1500 python tools/tune_sweep.py --repeat 2 base att_pf=RFE_ATT_PF=1 gemm_pf=RFE_GEMM_PF=1 conv_pad=RFE_CONV_PAD=1 att_dbuf=RFE_ATT_DBUF=1 all=RFE_ATT_PF=1,RFE_GEMM_PF=1,RFE_CONV_PAD=1 2>&1 | tee $OUT/sweep.txt
