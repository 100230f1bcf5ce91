#!/bin/bash
# GPU run B of round 2: GPU test suite (all failures listed), tolerance study on 40 cases
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02b
mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -8 $OUT/pytest.log
timeout 1500 python tools/lg_tolerance_study.py --cases 40 > $OUT/lg_tolerance.md 2> $OUT/lg_tolerance.err; echo "study rc=$?"
tail -4 $OUT/lg_tolerance.md
