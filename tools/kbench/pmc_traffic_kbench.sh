#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes over a kbench binary (GPU box only): usage tools/kbench/pmc_traffic_kbench.sh <tag> <binary> [args]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
BIN=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/f -o f -- $BIN "$@" > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/w -o w -- $BIN "$@" > $OUT/w.log 2>&1
cd $R
python3 tools/rocpd_pmc.py $(find $OUT -name "*_results.db") 2>&1 | head -30 > $OUT/traffic.md
cut -c1-220 $OUT/traffic.md | head -14
