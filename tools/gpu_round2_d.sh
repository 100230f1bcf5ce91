#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02d
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -4 $OUT/pytest.log
timeout 1500 python tools/tune_sweep.py --repeat 2 base rope_gemm=RFE_ROPE_IN_GEMM=1 rinit=RFE_GEMM_RINIT=1 rinit_pf=RFE_GEMM_RINIT=1,RFE_GEMM_PF=1 2>&1 | tee $OUT/sweep.txt
