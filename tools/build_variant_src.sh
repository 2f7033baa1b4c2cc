#!/bin/bash
# Like build_variant.sh, but the replaced object is compiled from an arbitrary source path (an experimental copy under build_ab/):
#   tools/build_variant_src.sh NAME kernels_update_mfma path/to/copy.hip "-DFLAGS"
set -e
cd "$(dirname "$0")/.."
NAME=$1; STEM=$2; SRC=$3; EXTRA=$4
C=ppo-libtorch_amd/csrc
mkdir -p build_ab
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Iinclude -I$C -Wall -Wno-unused-function"
case "$STEM" in kernels_update_mfma|kernels_gemm|kernels_generic_fused) FLAGS="$FLAGS -fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc $FLAGS $EXTRA -c $SRC -o build_ab/${STEM}_$NAME.o
OBJS=""
for f in api kernels_rollout kernels_gae kernels_update kernels_update_mfma kernels_generic kernels_generic_fused kernels_generic_bwd kernels_gemm; do
    if [ "$f" == "$STEM" ]; then OBJS="$OBJS build_ab/${f}_$NAME.o"; else OBJS="$OBJS $C/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/libppo_hip_$NAME.so $OBJS -ldl
echo build_ab/libppo_hip_$NAME.so
