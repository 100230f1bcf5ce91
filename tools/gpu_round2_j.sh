#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02j
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_stereo.py -m gpu -q 2>&1 | tail -4
timeout 1500 python tools/tune_sweep.py --repeat 3 base lni0=RFE_LN_INTERLEAVE=0 2>&1 | tee $OUT/sweep.txt | cut -c1-330
