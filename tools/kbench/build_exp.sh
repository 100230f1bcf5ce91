#!/bin/bash
# builds experiment variants of the tuning library: rover-slam_amd/exp/librover_fe_exp<N>.so (compile-time switches RFE_EXP=N in the two latency kernels)
set -e
cd /root/repo/rover-slam_amd/csrc
mkdir -p ../exp
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -DRFE_TUNING"
for N in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DRFE_EXP=$N -c gemm_lat.hip -o /tmp/gemm_lat.exp$N.o &
  /opt/rocm/bin/hipcc $FLAGS -DRFE_EXP=$N -c lg_attention_lat.hip -o /tmp/lg_attention_lat.exp$N.o &
done
wait
for N in "$@"; do
  OBJS=$(ls *.tuning.o | grep -v "gemm_lat.tuning.o\|lg_attention_lat.tuning.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../exp/librover_fe_exp$N.so $OBJS /tmp/gemm_lat.exp$N.o /tmp/lg_attention_lat.exp$N.o -ldl -pthread
done
ls -la ../exp
