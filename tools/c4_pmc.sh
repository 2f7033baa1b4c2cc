#!/bin/bash
# Counters of configs[4]'s share, one rocprofv3 --pmc pass per set (never beside a trace), summarised per kernel:  tools/c4_pmc.sh TAG ["SET" ...]
# With the nets on one stream (GEN_AB_ONE_STREAM builds) every kernel is alone on the chip and its counters are its own.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$(pwd); export TMPDIR=/tmp
TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
if [ $# -eq 0 ]; then set -- "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD"; fi
j=0
for set in "$@"; do
    j=$((j + 1))
    timeout -k 10 240 rocprofv3 --pmc $set -f csv -d "$OUT/pmc$j" -o run -- python3 $ROOT/tools/config4_bench.py > "$OUT/pmc$j.log" 2>&1 || echo "pmc pass $j FAILED"
    echo "pass $j ($set) done"
done
python3 $ROOT/tools/pmc_summary.py "$OUT/pmc_per_dispatch.json" "$OUT"/pmc* > "$OUT/pmc_summary.txt"
python3 - "$OUT/pmc_per_dispatch.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if "bwd_layer" in k or "gemm_kernel" in k or "generic_forward" in k:
        print(k)
        print("   ", {a: round(b, 1) for a, b in sorted(v.items())})
PY
