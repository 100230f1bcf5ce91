#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02e
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -6 $OUT/pytest.log
timeout 900 python tools/tune_sweep.py --repeat 2 base rope0=RFE_ROPE_MODE=0 rope1=RFE_ROPE_MODE=1 nopf=RFE_GEMM_PF=0 2>&1 | tee $OUT/sweep.txt | cut -c1-200
for i in 1 2; do timeout 300 python bench.py --workload c5 --steps 100 --warmup 10 > $OUT/bench_c5_$i.json 2> $OUT/bench_c5.err; cut -c1-330 $OUT/bench_c5_$i.json; done
timeout 300 python bench.py --workload c3 --steps 100 --warmup 10 | cut -c1-300
timeout 300 python bench.py --workload c2 --steps 100 --warmup 10 | cut -c1-300
