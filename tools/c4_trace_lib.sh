#!/bin/bash
# tools/c4_trace_lib.sh LIB.so : bench + kernel trace of configs[4]'s share with the given build of libppo_hip.so
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
cp ppo-libtorch_amd/libppo_hip.so /tmp/libppo_hip_orig.so
cp "$1" ppo-libtorch_amd/libppo_hip.so
bash tools/c4_trace.sh notest
cp /tmp/libppo_hip_orig.so ppo-libtorch_amd/libppo_hip.so
