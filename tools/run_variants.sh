export TMPDIR=/tmp
for v in abl_base abl_noe0 abl_noe1 abl_noe2 abl_nog; do
export PPO_HIP_LIBRARY=$PWD/build_ab/libppo_hip_$v.so
rm -rf gpurun_out/pm; timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -f csv -d gpurun_out/pm -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile 0 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os,collections
fs=glob.glob("gpurun_out/pm/**/*counter_collection.csv",recursive=True)
if not fs: print(os.environ["PPO_HIP_LIBRARY"][-12:], "no output"); raise SystemExit
acc=collections.defaultdict(float); n=collections.Counter()
for r in csv.DictReader(open(fs[0])):
    if "fwd_bwd_mfma_ws" in r["Kernel_Name"]:
        acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
print(os.environ["PPO_HIP_LIBRARY"][-12:], {k: round(v/n[k]) for k,v in acc.items()}, n.most_common(1))
PY
done
