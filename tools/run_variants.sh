export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_generic.py tests/test_gpu_config4_ref.py -m gpu -x -q 2>&1 | tail -3 || exit 1
rm -rf gpurun_out/kt; rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/kt -o run -- python3 tools/config4_bench.py > /dev/null 2>&1
python3 tools/c4_timeline.py $(find gpurun_out/kt -name '*kernel_trace.csv' | head -1) | head -3
for i in 1 2; do for v in build_ab/libppo_hip_head.so ppo-libtorch_amd/libppo_hip.so; do
PPO_HIP_LIBRARY=$PWD/$v timeout -k 10 200 python3 tools/config4_bench.py 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', round(d['env_steps_per_s']/1e6,3), 'M', round(d['minibatch_step_ms'],4), round(d['update_ms_per_step'],4), round(d['rollout_ms'],3), d['loss'])"
done; done
