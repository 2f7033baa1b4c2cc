#!/bin/bash
# tools/probe_ab.sh lib_A.so lib_B.so ...: the critic-pass probe (tools/fused_fwd_probe.py) under each build, one gpurun call
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
cp ppo-libtorch_amd/libppo_hip.so /tmp/libppo_hip_orig.so
for src in "$@"; do
    cp "$src" ppo-libtorch_amd/libppo_hip.so
    echo "$src $(python3 tools/fused_fwd_probe.py 2>/dev/null | tail -n 1)"
done
cp /tmp/libppo_hip_orig.so ppo-libtorch_amd/libppo_hip.so
