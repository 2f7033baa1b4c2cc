#!/bin/bash
# GAE size sweep of several builds inside one gpurun call: tools/sweep_ab.sh "SIZES" lib_A.so lib_B.so ...
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
SIZES=$1; shift
for r in 1 2; do
for src in "$@"; do
    export PPO_HIP_LIBRARY="$(realpath "$src")"   # binding.py loads this build; the shipped library is never overwritten
    echo "== $src"
    python3 tools/gae_sweep.py $SIZES 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['N'], round(d['us'], 2), 'us', round(d['frac_of_8TBps'], 3), d['bit_exact_vs_oracle'])"
done
done
