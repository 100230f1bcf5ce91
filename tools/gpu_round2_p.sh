#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02p
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for w in c2 c3 c5; do timeout 300 python bench.py --workload $w --steps 100 --warmup 10 > $OUT/lat_$w.json 2>> $OUT/err.log; cut -c1-170 $OUT/lat_$w.json; done
timeout 600 python tools/tune_sweep.py --repeat 2 base 2>&1 | tee $OUT/sweep.txt | cut -c1-200
