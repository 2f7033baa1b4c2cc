#!/bin/bash
# phase stamps of several builds inside one gpurun call: tools/stamps_ab.sh lib_A.so lib_B.so ...
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
cp ppo-libtorch_amd/libppo_hip.so /tmp/libppo_hip_orig.so
for src in "$@"; do
    cp "$src" ppo-libtorch_amd/libppo_hip.so
    echo "== $src"
    python3 tools/phase_stamps.py 2>&1 | tail -n 3
done
cp /tmp/libppo_hip_orig.so ppo-libtorch_amd/libppo_hip.so
