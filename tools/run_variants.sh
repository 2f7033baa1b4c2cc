export TMPDIR=/tmp
PPO_HIP_LIBRARY=$PWD/build_ab/libppo_hip_epikeep.so timeout -k 10 900 python3 -m pytest tests/test_gpu_config4_ref.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do for v in ppo-libtorch_amd/libppo_hip.so build_ab/libppo_hip_epikeep.so; do
PPO_HIP_LIBRARY=$PWD/$v timeout -k 10 200 python3 tools/config4_bench.py 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', round(d['env_steps_per_s']/1e6,3), 'M', round(d['minibatch_step_ms'],4), round(d['update_ms_per_step'],4), round(d['rollout_ms'],3), d['loss'])"
done; done
