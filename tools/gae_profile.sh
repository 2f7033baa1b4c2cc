#!/bin/bash
# The GAE scan by itself on the GPU box:  tools/gae_profile.sh r03_v1
#   in-trace durations by size (kernel trace), memory-side traffic by size (two --pmc passes), back-to-back wall times (gae_sweep's own lines)
set -e -o pipefail
TAG=${1:-r03}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT" "$ROOT/profiles"
SIZES="4096 8192 32768"
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/gae_trace" -o run -- python3 $ROOT/tools/gae_sweep.py $SIZES 131072 > "$OUT/gae_sweep.jsonl" 2> "$OUT/gae_trace.log"
grep '^{' "$OUT/gae_sweep.jsonl" > "$ROOT/profiles/${TAG}_gae_sweep.jsonl" || true
python3 $ROOT/tools/gae_by_size.py "$(find "$OUT/gae_trace" -name '*kernel_trace.csv' | head -n 1)" "$ROOT/profiles/${TAG}_gae_by_size.json" > "$OUT/gae_by_size.txt"
rocprofv3 --pmc FETCH_SIZE -f csv -d "$OUT/gae_pmc1" -o run -- python3 $ROOT/tools/gae_sweep.py $SIZES > "$OUT/gae_pmc1.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d "$OUT/gae_pmc2" -o run -- python3 $ROOT/tools/gae_sweep.py $SIZES > "$OUT/gae_pmc2.log" 2>&1
python3 $ROOT/tools/gae_traffic.py "$ROOT/profiles/${TAG}_gae_traffic.json" "$OUT/gae_pmc1" "$OUT/gae_pmc2" > "$OUT/gae_traffic.txt"
cat "$OUT/gae_by_size.txt" "$OUT/gae_traffic.txt"
cp "$ROOT"/profiles/${TAG}_gae_* "$OUT/"
