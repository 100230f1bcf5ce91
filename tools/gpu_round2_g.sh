#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02g
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
grep -E "^E  |passed|failed" $OUT/pytest.log | head -20 | cut -c1-300
timeout 900 python tools/tune_sweep.py --repeat 3 base lnfuse0=RFE_LN_FUSE=0 2>&1 | tee $OUT/sweep.txt | cut -c1-330
