#!/bin/bash
# Runs on the MI355X box (via gpurun, after `python tools/stamp.py` in the build container): plain bench, rocprofv3 kernel trace, the PMC passes
# (each in its own run, kernel-trace only -- see the task's rocprofv3 rules), the latency workloads with their own kernel traces, the
# "what the waves wait on" passes, and the same set for the RFE_OPT_LG_FP16X2 diagnostic configuration.  Outputs -> gpurun_out/<tag>/
# usage: tools/profile_round.sh <tag> [quick]      (quick: default configuration only)
set -u
TAG=${1:-r01}
QUICK=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python3 tools/stamp.py --box $OUT/stamp.json
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
tail -c 400 $OUT/bench.err
B="--no-cpu-baseline --no-variants --no-pool --no-latency --sustained-steps 0"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 $R/bench.py --steps 3 --warmup 1 $B > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o f -- python3 $R/bench.py --steps 2 --warmup 1 $B > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o w -- python3 $R/bench.py --steps 2 --warmup 1 $B > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/pmc_sq -o s -- python3 $R/bench.py --steps 2 --warmup 1 $B > $OUT/pmc_sq.log 2>&1
cd $R
for w in c2 c3 c5; do
  python bench.py --workload $w --steps 100 --warmup 10 > $OUT/lat_$w.json 2>> $OUT/bench.err
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/trace_$w -o t -- python3 $R/bench.py --workload $w --steps 30 --warmup 5 > $OUT/trace_$w.log 2>&1)
  python3 tools/rocpd_stats.py $(find $OUT/trace_$w -name "*_results.db" | head -1) --gaps > $OUT/stats_$w.md
  rm -rf $OUT/trace_$w
done
bash tools/pmc_wait.sh ${TAG}_wait --steps 2 --warmup 1 > /dev/null 2>&1
bash tools/pmc_wait.sh ${TAG}_wait_c3 --workload c3 --steps 10 --warmup 2 > /dev/null 2>&1
if [ -z "$QUICK" ]; then
  # RFE_OPT_LG_FP16X2 = 1 (default off; never the headline): the option's own evidence, from the same binary
  python bench.py --steps 20 --warmup 3 --lg-fp16x2 1 --no-pool --no-pcie > $OUT/bench_fp16x2.json 2>> $OUT/bench.err
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/trace_fp16x2 -o t -- python3 $R/bench.py --steps 3 --warmup 1 $B --no-pcie --lg-fp16x2 1 > $OUT/trace_fp16x2.log 2>&1)
  python3 tools/rocpd_stats.py $(find $OUT/trace_fp16x2 -name "*_results.db" | head -1) > $OUT/stats_fp16x2.md
  (cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $OUT/pmc_sq_fp16x2 -o s -- python3 $R/bench.py --steps 2 --warmup 1 $B --no-pcie --lg-fp16x2 1 > $OUT/pmc_sq_fp16x2.log 2>&1)
  python3 tools/rocpd_pmc.py $(find $OUT/pmc_sq_fp16x2 -name "*_results.db" | head -1) > $OUT/pmc_sq_fp16x2.md
  rm -rf $OUT/trace_fp16x2 $OUT/pmc_sq_fp16x2
  bash tools/pmc_wait.sh ${TAG}_wait_fp16x2 --steps 2 --warmup 1 --lg-fp16x2 1 > /dev/null 2>&1
  # ... and at the reference's own shape: one pair per call with the option on (split forms of the latency kernels)
  for w in c3 c5; do
    python bench.py --workload $w --steps 100 --warmup 10 --lg-fp16x2 1 > $OUT/lat_${w}_fp16x2.json 2>> $OUT/bench.err
  done
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/trace_c3_fp16x2 -o t -- python3 $R/bench.py --workload c3 --steps 30 --warmup 5 --lg-fp16x2 1 > $OUT/trace_c3_fp16x2.log 2>&1)
  python3 tools/rocpd_stats.py $(find $OUT/trace_c3_fp16x2 -name "*_results.db" | head -1) --gaps > $OUT/stats_c3_fp16x2.md
  rm -rf $OUT/trace_c3_fp16x2
fi
# the summaries are written HERE, on the box (the rocpd databases are far beyond the 64 MiB gpurun merges back): profiles/<rnd>_* -> gpurun_out/<tag>_profiles/
RND=${TAG:0:3}
python3 tools/refresh_profiles.py $TAG $RND > $OUT/refresh.log 2>&1
mkdir -p $R/gpurun_out/${TAG}_profiles
cp $R/profiles/${RND}_* $R/gpurun_out/${TAG}_profiles/ 2>/dev/null
cp $OUT/*.json $OUT/*.md $OUT/*.log $OUT/*.err $R/gpurun_out/${TAG}_profiles/ 2>/dev/null
find $R/gpurun_out -name "*.db" -delete
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
tail -5 $OUT/refresh.log
du -sh $R/gpurun_out
cat $OUT/bench.json | cut -c1-600
