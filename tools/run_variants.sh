export TMPDIR=/tmp
for v in ppo-libtorch_amd/libppo_hip.so build_ab/libppo_hip_noobs.so build_ab/libppo_hip_nosample.so build_ab/libppo_hip_noboth.so; do
PPO_HIP_LIBRARY=$PWD/$v timeout -k 10 200 python3 tools/config4_bench.py 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', round(d['rollout_ms'],3))"
done
