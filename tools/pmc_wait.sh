#!/bin/bash
# PMC passes for "what do the waves wait on" (wave-cycle breakdown per kernel); GPU box only.  usage: tools/pmc_wait.sh <tag> [bench args]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $OUT/w1 -o a -- python3 $R/bench.py "$@" --no-cpu-baseline --no-variants --no-pool --no-pcie --no-latency --sustained-steps 0 > $OUT/w1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM --kernel-trace -d $OUT/w2 -o b -- python3 $R/bench.py "$@" --no-cpu-baseline --no-variants --no-pool --no-pcie --no-latency --sustained-steps 0 > $OUT/w2.log 2>&1
cd $R && python3 tools/pmc_wait_table.py --table $TAG
find $OUT -name "*_results.db" -size +20M -delete    # large traces stay on the box: gpurun merges at most 64 MiB back
tail -3 $OUT/table.md | cut -c1-200
