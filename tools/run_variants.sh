PPO_HIP_LIBRARY=$PWD/ppo-libtorch_amd/libppo_hip.so python tools/bwd_check.py gpurun_out/g_fused.npz
timeout -k 10 600 python -m pytest tests/test_gpu_generic.py tests/test_gpu_config4_ref.py -q -x 2>&1 | tail -4 | cut -c1-300
tools/c4_ab.sh 3 build_ab/libppo_hip_sephead.so ppo-libtorch_amd/libppo_hip.so
