#!/bin/bash
# A/B(/C...) timing of several builds of libppo_hip.so inside ONE gpurun call (boxes differ by a few percent between calls):
#   tools/ab.sh ROUNDS lib_A.so lib_B.so [lib_C.so ...]
# Alternates the builds round-robin (selected through PPO_HIP_LIBRARY: the shipped library is never overwritten) and prints, for each
# run, env-steps/s, ms per iteration and the HIP-event times of the update kernel (per launch), of rollout + critic batch, and of the
# optimizer step (every kernel bracketed: --profile 1).
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$1; shift
for i in $(seq 1 $R); do
    for src in "$@"; do
        PPO_HIP_LIBRARY="$(realpath "$src")" python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads --repeats 0 --profile 1 2>/dev/null | tail -n 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-44s' % '$src', round(d['value']/1e6,2), 'M env-steps/s', round(d['ms_per_step'],3), 'ms', round(1e3*d['roofline']['avg_launch_ms'],2), 'us/update launch', round(1e3*d['phase_ms_per_step']['rollout'],1), 'us rollout+values', round(25*((d['phase_ms_per_step']['clip_adamw'] or 0)+(d['phase_ms_per_step']['grad_reduce'] or 0)),2), 'us/optimizer step')"
    done
done
