#!/bin/bash
# Regenerates the judged profile summaries of one round on the GPU box:  tools/collect_profiles.sh r01_v3
#   gpurun_out/<tag>/...  raw rocprofv3 output (scratch)        profiles/<tag>_*  summaries (copy these back and commit)
# kernel-trace/stats and every --pmc pass are separate runs of the same bench command (counters never share a run with a trace).
set -e -o pipefail
TAG=${1:-r01}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT" "$ROOT/profiles"
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline"

python3 $ROOT/bench.py --steps 20 --warmup 3 --profile 2 > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -n 1 "$OUT/bench.json" > "$ROOT/profiles/${TAG}_bench.json"

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
cp "$ROOT"/profiles/${TAG}_* "$OUT/"
echo done
