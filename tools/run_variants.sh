export TMPDIR=/tmp
for v in build_ab/libppo_hip_r4.so ppo-libtorch_amd/libppo_hip.so; do
  export PPO_HIP_LIBRARY=$PWD/$v
  rm -rf gpurun_out/kt; rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/kt -o run -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --profile 0 > /dev/null 2>&1
  echo "== $v"
  python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/kt/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print(r["Name"][:70].replace("(anonymous namespace)::",""), r["Calls"], round(float(r["AverageNs"])/1e3,2))
PY
done
