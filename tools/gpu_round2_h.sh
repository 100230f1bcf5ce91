#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02h
mkdir -p $OUT
cd $R
timeout 1500 python tools/tune_sweep.py --repeat 2 base gemm_stag=RFE_GEMM_STAGGER=1 att_stag=RFE_ATT_PRIO=2 att_noprio=RFE_ATT_PRIO=0 conv_stag=RFE_CONV_STAGGER=1 all=RFE_GEMM_STAGGER=1,RFE_ATT_PRIO=2,RFE_CONV_STAGGER=1 2>&1 | tee $OUT/sweep.txt | cut -c1-330
