#!/bin/bash
# Runs on the MI355X box (via gpurun): plain bench, rocprofv3 kernel trace, and the PMC passes
# (each in its own run, kernel-trace only -- see the task's rocprofv3 rules).  Outputs -> gpurun_out/<tag>/
# usage: tools/profile_round.sh <tag>
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
tail -c 400 $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-pool --sustained-steps 0 > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --no-pool --sustained-steps 0 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --no-pool --sustained-steps 0 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/pmc_sq -o s -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --no-pool --sustained-steps 0 > $OUT/pmc_sq.log 2>&1
cd $R
for w in c2 c3 c5; do python bench.py --workload $w --steps 100 --warmup 10 > $OUT/lat_$w.json 2>> $OUT/bench.err; done
find $OUT -name "*_results.db" | head
cat $OUT/bench.json | cut -c1-1500
