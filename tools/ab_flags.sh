#!/bin/bash
# A/B of ppo_config.kernel_flags settings on the SAME library inside one gpurun call:  tools/ab_flags.sh ROUNDS flagsA flagsB ...
# (include/ppo_hip.h PPO_KERNEL_*: 0 = defaults, 1 vector rollout, 2 vector update, 4 one-wave matrix-core update).  Prints what tools/ab.sh prints.
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$1; shift
for i in $(seq 1 $R); do
    for val in "$@"; do
        python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --profile 1 --kernel-flags $val 2>/dev/null | tail -n 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-16s' % 'kernel_flags=$val', round(d['value']/1e6,2), 'M env-steps/s', round(d['ms_per_step'],3), 'ms', round(1e3*d['roofline']['avg_launch_ms'],2), 'us/update launch', round(1e3*d['phase_ms_per_step']['rollout'],1), 'us rollout+values', round(25*((d['phase_ms_per_step']['clip_adamw'] or 0)+(d['phase_ms_per_step']['grad_reduce'] or 0)),2), 'us/optimizer step')"
    done
done
