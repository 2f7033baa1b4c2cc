#!/bin/bash
# phase stamps of several builds inside one gpurun call: tools/stamps_ab.sh lib_A.so lib_B.so ...
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for src in "$@"; do
    export PPO_HIP_LIBRARY="$(realpath "$src")"   # binding.py loads this build; the shipped library is never overwritten
    echo "== $src"
    python3 tools/phase_stamps.py 2>&1 | tail -n 3
done
