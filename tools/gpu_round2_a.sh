#!/bin/bash
# GPU run A of round 2: full GPU test suite, default bench, 2-rank gloo functional check, stereo stream, tolerance study
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02a
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cut -c1-600 $OUT/bench.json
RFE_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --sustained-steps 20 > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err; echo "gloo2 rc=$?"
cut -c1-300 $OUT/bench_gloo2.json; tail -3 $OUT/bench_gloo2.err
timeout 300 python bench.py --workload c5 --steps 50 --warmup 5 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; echo "c5 rc=$?"
cat $OUT/bench_c5.json | cut -c1-600; tail -3 $OUT/bench_c5.err
timeout 300 python bench.py --lg-fold 1 --no-cpu-baseline --no-pcie --sustained-steps 0 > $OUT/bench_fold1.json 2> $OUT/bench_fold1.err
cut -c1-200 $OUT/bench_fold1.json
timeout 900 python tools/lg_tolerance_study.py --cases 20 > $OUT/lg_tolerance.md 2> $OUT/lg_tolerance.err; echo "study rc=$?"
tail -25 $OUT/lg_tolerance.md
