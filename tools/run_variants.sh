export TMPDIR=/tmp
for v in ppo-libtorch_amd/libppo_hip.so build_ab/libppo_hip_epb64.so; do
echo "== $v"
PPO_HIP_LIBRARY=$PWD/$v timeout -k 10 300 python3 tools/gae_sweep.py 131072 1048576 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print({k:d[k] for k in d if k in ('envs','N','GBps','algorithmic_GBps','us','frac','time_us','GB_per_s','ok','checked')} or l[:200])"
done
