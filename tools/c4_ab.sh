#!/bin/bash
# A/B of libppo_hip.so builds on configs[4]'s share inside ONE gpurun call:  tools/c4_ab.sh ROUNDS lib_A.so lib_B.so ...
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$1; shift
for i in $(seq 1 $R); do
    for src in "$@"; do
        export PPO_HIP_LIBRARY="$(realpath "$src")"   # binding.py loads this build; the shipped library is never overwritten
        python3 tools/config4_bench.py 2>/dev/null | tail -n 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-40s' % '$src', round(d['env_steps_per_s']/1e6,3), 'M env-steps/s', round(d['minibatch_step_ms'],4), 'ms step', round(d['update_ms_per_step'],4), 'ms all-in')"
    done
done
