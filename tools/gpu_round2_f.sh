#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02f
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -4 $OUT/pytest.log
timeout 900 python tools/tune_sweep.py --repeat 3 base 2>&1 | tee $OUT/sweep.txt | cut -c1-200
timeout 300 python bench.py --workload c5 --steps 100 --warmup 10 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; cut -c1-330 $OUT/bench_c5.json
