#!/bin/bash
# PMC passes over tools/matmul_pmc.py, summarised:  tools/matmul_pmc.sh TAG [K] [prec]
set -e -o pipefail
TAG=$1; K=${2:-1024}; PREC=${3:-0}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$(pwd); export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM"; do
    i=$((i + 1))
    rocprofv3 --pmc $set -f csv -d "$OUT/pmc$i" -o run -- python3 $ROOT/tools/matmul_pmc.py $K $PREC > "$OUT/pmc$i.log" 2>&1 || echo "pass $i failed"
done
python3 $ROOT/tools/pmc_summary.py "$OUT/summary.json" "$OUT"/pmc* > "$OUT/summary.txt"
python3 - "$OUT/summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if k.startswith("gemm"):
        print(k)
        for a, b in sorted(v.items()):
            print("    %-28s %14.1f" % (a, b))
PY
