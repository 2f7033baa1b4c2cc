#!/bin/bash
# tools/probe_ab.sh lib_A.so lib_B.so ...: the critic-pass probe (tools/fused_fwd_probe.py) under each build, one gpurun call
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for src in "$@"; do
    export PPO_HIP_LIBRARY="$(realpath "$src")"   # binding.py loads this build; the shipped library is never overwritten
    echo "$src $(python3 tools/fused_fwd_probe.py 2>/dev/null | tail -n 1)"
done
