#!/bin/bash
# Builds a variant of libppo_hip.so for an A/B inside one gpurun call:  tools/build_variant.sh NAME FILE.hip "-DSOME_FLAG=1 ..."
#   -> build_ab/libppo_hip_NAME.so  (FILE.hip recompiled with the extra flags, every other object taken from the current build)
set -e
cd "$(dirname "$0")/.."
NAME=$1; FILE=$2; EXTRA=$3
C=ppo-libtorch_amd/csrc
mkdir -p build_ab
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Iinclude -I$C -Wall -Wno-unused-function"
case "$FILE" in kernels_update_mfma.hip|kernels_gemm.hip|kernels_generic_fused.hip|kernels_generic_bwd.hip) FLAGS="$FLAGS -fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc $FLAGS $EXTRA -c $C/$FILE -o build_ab/${FILE%.hip}_$NAME.o
OBJS=""
for f in api kernels_rollout kernels_gae kernels_update kernels_update_mfma kernels_generic kernels_generic_fused kernels_generic_bwd kernels_gemm; do
    if [ "$f.hip" == "$FILE" ]; then OBJS="$OBJS build_ab/${f}_$NAME.o"; else OBJS="$OBJS $C/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/libppo_hip_$NAME.so $OBJS -ldl
echo build_ab/libppo_hip_$NAME.so
