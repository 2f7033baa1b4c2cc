#!/bin/bash
# A/B of several builds of libppo_hip.so on the GEMM benchmark inside ONE gpurun call:  tools/ab_mm.sh lib_A.so lib_B.so ...
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for src in "$@"; do
    export PPO_HIP_LIBRARY="$(realpath "$src")"   # binding.py loads this build; the shipped library is never overwritten
    echo "== $src"
    python3 tools/matmul_bench.py 2>/dev/null | grep -v "d(weight)\|head" | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('   %-22s %-6s %7.1f us %7.1f TF' % (d['case'], d['precision'], d['us'], d['tflops']))"
done
