#!/bin/bash
# One rocprofv3 --pmc pass over a short bench run, summarised per kernel:  tools/pmc_quick.sh TAG "COUNTER COUNTER ..."
set -e -o pipefail
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$(pwd); export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
i=0
for set in "$@"; do
    i=$((i + 1))
    rocprofv3 --pmc $set -f csv -d "$OUT/pmc$i" -o run -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile 0 > "$OUT/pmc$i.log" 2>&1 || echo "pass $i failed"
done
python3 $ROOT/tools/pmc_summary.py "$OUT/summary.json" "$OUT"/pmc* > "$OUT/summary.txt"
python3 - "$OUT/summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if k.startswith("fwd_bwd") or k.startswith("rollout") or k.startswith("values"):
        print(k)
        for a, b in sorted(v.items()):
            print("    %-28s %14.1f" % (a, b))
PY
