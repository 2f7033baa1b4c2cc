export TMPDIR=/tmp
for v in ppo-libtorch_amd/libppo_hip.so build_ab/libppo_hip_parts64.so build_ab/libppo_hip_parts128.so; do
export PPO_HIP_LIBRARY=$PWD/$v
rm -rf gpurun_out/kt; rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/kt -o run -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --profile 0 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os
f=glob.glob("gpurun_out/kt/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f))):
    if "perm_adv" in r["Name"] or "adv_norm" in r["Name"] or "fillBuffer" in r["Name"]: print(os.environ["PPO_HIP_LIBRARY"][-14:], r["Name"][:40].replace("(anonymous namespace)::",""), r["Calls"], round(float(r["AverageNs"])/1e3,2))
PY
done
unset PPO_HIP_LIBRARY
for i in 1 2; do for v in ppo-libtorch_amd/libppo_hip.so build_ab/libppo_hip_parts64.so build_ab/libppo_hip_parts128.so; do PPO_HIP_LIBRARY=$PWD/$v python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --profile 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', d['ms_per_step'], round(d['value']/1e6,2))"; done; done
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
