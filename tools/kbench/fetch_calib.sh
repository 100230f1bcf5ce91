#!/bin/bash
# GPU box only: FETCH_SIZE / WRITE_SIZE of the calibration kernels (separate passes, kernel-trace only).  usage: tools/kbench/fetch_calib.sh <tag>
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/f -o f -- $R/tools/kbench/fetch_calib > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/w -o w -- $R/tools/kbench/fetch_calib > $OUT/w.log 2>&1
cd $R
python3 tools/rocpd_pmc.py $(find $OUT -name "*_results.db") 2>&1 | head -12 > $OUT/calib.md
cat $OUT/calib.md | cut -c1-200; tail -1 $OUT/f.log
