#!/bin/bash
# PMC passes over a kbench binary (GPU box only): usage tools/kbench/pmc_kbench.sh <tag> <binary> [args]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
BIN=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $OUT/w1 -o a -- $BIN "$@" > $OUT/w1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/w2 -o b -- $BIN "$@" > $OUT/w2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU --kernel-trace -d $OUT/w3 -o c -- $BIN "$@" > $OUT/w3.log 2>&1
cd $R
python3 tools/rocpd_pmc.py $(find $OUT -name "*_results.db") 2>&1 | head -40 > $OUT/pmc.md
cat $OUT/pmc.md | cut -c1-400 | head -12
