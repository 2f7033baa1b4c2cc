#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for src in "$@"; do
  export PPO_HIP_LIBRARY="$(realpath "$src")"
  for set in FETCH_SIZE WRITE_SIZE; do
    D=/tmp/pf_$(basename $src)_$set; rm -rf $D
    rocprofv3 --pmc $set -f csv -d $D -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads --repeats 0 > $D.out 2>&1
  done
  python3 tools/pmc_summary.py /tmp/pf_$(basename $src).json /tmp/pf_$(basename $src)_FETCH_SIZE /tmp/pf_$(basename $src)_WRITE_SIZE > /dev/null
  python3 - "$src" /tmp/pf_$(basename $src).json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
for k, v in d.items():
    if "fwd_bwd_mfma_ws" in k or "pack_records" in k:
        print(sys.argv[1], k, {a: round(b / 1e6, 2) for a, b in v.items() if a.startswith("hbm")})
PY
done
