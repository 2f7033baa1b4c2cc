#!/bin/bash
# tools/c4_trace_lib.sh LIB.so : bench + kernel trace of configs[4]'s share with the given build of libppo_hip.so
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export PPO_HIP_LIBRARY="$(realpath "$1")"   # binding.py loads this build; the shipped library is never overwritten
bash tools/c4_trace.sh notest
