#!/bin/bash
# A/B of several builds of libppo_hip.so on the headline workload WITHOUT per-kernel events (tools/ab.sh brackets every kernel):
#   tools/ab_plain.sh ROUNDS lib_A.so lib_B.so ...   -> env-steps/s and ms per iteration of `bench.py --steps 100`, builds alternating
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$1; shift
for i in $(seq 1 $R); do
    for src in "$@"; do
        export PPO_HIP_LIBRARY="$(realpath "$src")"   # binding.py loads this build; the shipped library is never overwritten
        python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --profile 0 ${WORKLOAD:+--workload $WORKLOAD} 2>/dev/null | tail -n 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-36s' % '$src', round(d['value']/1e6,2), 'M env-steps/s', round(d['ms_per_step'],4), 'ms')"
    done
done
