#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02m
mkdir -p $OUT
cd $R
timeout 500 python tools/fuzz_parity.py 240 101 > $OUT/fuzz1.log 2>&1; tail -3 $OUT/fuzz1.log
timeout 500 python tools/fuzz_parity.py 240 202 > $OUT/fuzz2.log 2>&1; tail -3 $OUT/fuzz2.log
timeout 300 python tools/fuzz_threads.py > $OUT/fuzz_threads.log 2>&1; tail -3 $OUT/fuzz_threads.log
