#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02l
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -3 $OUT/pytest.log
timeout 900 python tools/tune_sweep.py --repeat 3 base 2>&1 | tee $OUT/sweep.txt | cut -c1-330
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --sustained-steps 0 > $OUT/pmc_fetch.log 2>&1
cd $R; python tools/rocpd_pmc.py $OUT/pmc_fetch/*/f_results.db 2>/dev/null | grep -E "conv|kernel \|" | cut -c1-160 || python tools/rocpd_pmc.py $OUT/pmc_fetch/f_results.db | grep -E "conv|kernel \|" | cut -c1-160
