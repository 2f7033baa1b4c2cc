#!/bin/bash
# build_all_variant.sh NAME "-DFLAGS": every object rebuilt with the flags -> build_ab/libppo_hip_NAME.so
set -e
cd "$(dirname "$0")/.."
NAME=$1; EXTRA=$2; C=ppo-libtorch_amd/csrc
OBJS=""
for f in api kernels_rollout kernels_gae kernels_update kernels_update_mfma kernels_generic kernels_generic_fused kernels_generic_bwd kernels_gemm; do
  FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Iinclude -I$C -Wall -Wno-unused-function"
  case "$f" in kernels_update_mfma|kernels_gemm|kernels_generic_fused|kernels_generic_bwd) FLAGS="$FLAGS -fno-slp-vectorize";; esac
  /opt/rocm/bin/hipcc $FLAGS $EXTRA -c $C/$f.hip -o build_ab/${f}_$NAME.o &
  OBJS="$OBJS build_ab/${f}_$NAME.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/libppo_hip_$NAME.so $OBJS -ldl
echo build_ab/libppo_hip_$NAME.so
