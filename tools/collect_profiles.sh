#!/bin/bash
# Regenerates the judged profile summaries of one round on the GPU box:  tools/collect_profiles.sh r01_v3
#   gpurun_out/<tag>/...  raw rocprofv3 output (scratch)        profiles/<tag>_*  summaries (copy these back and commit)
# kernel-trace/stats and every --pmc pass are separate runs of the same bench command (counters never share a run with a trace).
set -e -o pipefail
TAG=${1:-r01}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$(pwd)
export TMPDIR=/tmp
# raw rocprofv3 output goes to /tmp on the box (a set's traces pass the 64 MiB gpurun merges back, and then NOTHING comes back); the summaries and logs to gpurun_out/<tag>/
OUT=/tmp/collect_$TAG
KEEP=$ROOT/gpurun_out/$TAG
rm -rf "$OUT"
mkdir -p "$OUT" "$KEEP" "$ROOT/profiles"
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads --repeats 0"   # the headline workload alone under the tracer / the counters

rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace" -o run -- $BENCH > "$OUT/trace.log" 2>&1
cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -n 1)" "$ROOT/profiles/${TAG}_kernel_stats.csv"

i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU"; do
    i=$((i + 1))
    rocprofv3 --pmc $set -f csv -d "$OUT/pmc$i" -o run -- $BENCH > "$OUT/pmc$i.log" 2>&1 || echo "pmc pass $i FAILED (see pmc$i.log)"
    echo "pmc pass $i done"
done
python3 $ROOT/tools/pmc_summary.py "$ROOT/profiles/${TAG}_pmc_per_dispatch.json" "$OUT"/pmc* > "$OUT/pmc_summary.txt"

# the GAE scan by itself: kernel-only durations at 4096 / 8192 / 32768 envs (same trace holds the three sizes), and the floor probe -- a stream
# kernel and the scan's own strip decomposition without the chain -- under the same tracer
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/gae_trace" -o run -- python3 $ROOT/tools/gae_sweep.py 4096 8192 32768 131072 1048576 > "$OUT/gae_sweep.jsonl" 2> "$OUT/gae_trace.log"
cp "$(find "$OUT/gae_trace" -name '*kernel_stats.csv' | head -n 1)" "$ROOT/profiles/${TAG}_gae_kernel_stats.csv"
grep '^{' "$OUT/gae_sweep.jsonl" > "$ROOT/profiles/${TAG}_gae_sweep.jsonl" || true
python3 $ROOT/tools/gae_by_size.py "$(find "$OUT/gae_trace" -name '*kernel_trace.csv' | head -n 1)" "$ROOT/profiles/${TAG}_gae_by_size.json" > "$OUT/gae_by_size.txt"
if [ -x $ROOT/tools/probes/gae_floor ]; then
    $ROOT/tools/probes/gae_floor 4096 8192 32768 > "$ROOT/profiles/${TAG}_gae_floor.jsonl"
    rocprofv3 --kernel-trace --stats -f csv -d "$OUT/floor_trace" -o run -- $ROOT/tools/probes/gae_floor 4096 > "$OUT/floor_trace.log" 2>&1
    cp "$(find "$OUT/floor_trace" -name '*kernel_stats.csv' | head -n 1)" "$ROOT/profiles/${TAG}_gae_floor_kernel_stats.csv"
fi
# FETCH_SIZE calibration on the build's two read patterns (tools/pmc_summary.py)
if [ -x $ROOT/tools/probes/fetch_calib ]; then
    rocprofv3 --pmc FETCH_SIZE -f csv -d "$OUT/calib" -o run -- $ROOT/tools/probes/fetch_calib > "$OUT/calib.log" 2>&1
    python3 $ROOT/tools/pmc_summary.py "$ROOT/profiles/${TAG}_fetch_calib.json" "$OUT/calib" > "$OUT/calib_summary.txt"
fi
# one GPU's share of BASELINE configs[4] in bf16: kernel durations
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/c4_trace" -o run -- python3 $ROOT/tools/config4_bench.py > "$OUT/c4_bench.json" 2> "$OUT/c4_trace.log"
cp "$(find "$OUT/c4_trace" -name '*kernel_stats.csv' | head -n 1)" "$ROOT/profiles/${TAG}_config4_kernel_stats.csv"
tail -n 1 "$OUT/c4_bench.json" > "$ROOT/profiles/${TAG}_config4_bench.json"
# the same share under the memory counters (separate passes, counters only): HBM bytes per minibatch step of the generic path = the sum over its
# kernels of bytes per dispatch x dispatches per step (tools/c4_traffic.py) -> bench.py --workload config4 reports it as roofline.traffic
j=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
    j=$((j + 1))
    rocprofv3 --pmc $set -f csv -d "$OUT/c4_pmc$j" -o run -- python3 $ROOT/tools/config4_bench.py > "$OUT/c4_pmc$j.log" 2>&1 || echo "config4 pmc pass $j FAILED"
done
python3 $ROOT/tools/pmc_summary.py "$OUT/c4_pmc_per_dispatch.json" "$OUT"/c4_pmc* > "$OUT/c4_pmc_summary.txt"
python3 $ROOT/tools/c4_traffic.py "$OUT/c4_pmc_per_dispatch.json" "$ROOT/profiles/${TAG}_config4_kernel_stats.csv" > "$ROOT/profiles/${TAG}_config4_traffic.json"
# the wave-specialised update kernel's wait shares (diagnostic build, if present) and the learning-curve comparison
if [ -f $ROOT/build_ab/libppo_hip_ws_stamp.so ]; then PPO_HIP_LIBRARY=$ROOT/build_ab/libppo_hip_ws_stamp.so python3 $ROOT/tools/ws_stamps.py > "$ROOT/profiles/${TAG}_ws_stamps.txt" 2>&1 || true; fi
python3 $ROOT/tests/test_gpu_curves.py > "$ROOT/profiles/${TAG}_curves.json" 2> "$OUT/curves.err" || echo "curves FAILED"
# what ties this set to the binary: the fingerprint of the sources it was built from (bench.py quotes a set only while it matches)
python3 $ROOT/tools/src_fingerprint.py --meta "$TAG" > "$OUT/meta.txt"
# the judged bench lines LAST: they quote this set (traffic, in-trace durations, floor, curves), which exists and is tied only now
python3 $ROOT/bench.py --steps 20 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -n 1 "$OUT/bench.json" > "$ROOT/profiles/${TAG}_bench.json"
python3 $ROOT/bench.py --workload config4 --no-cpu-baseline > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err" && tail -n 1 "$OUT/bench_c4.json" > "$ROOT/profiles/${TAG}_bench_default_config4.json" || echo "config4 bench FAILED"
python3 $ROOT/bench.py --workload mountaincar --no-cpu-baseline > "$OUT/bench_mc.json" 2> "$OUT/bench_mc.err" && tail -n 1 "$OUT/bench_mc.json" > "$ROOT/profiles/${TAG}_bench_default_mountaincar.json" || echo "mountaincar bench FAILED"
cp "$ROOT"/profiles/${TAG}_* "$KEEP/"
cp "$OUT"/*.log "$OUT"/*.txt "$OUT"/*.err "$OUT"/*.json "$KEEP/" 2>/dev/null || true
echo done
