#!/bin/bash
# In-trace average durations of named kernels of configs[4]'s share for several builds inside ONE gpurun call:  tools/c4_kernel_ab.sh "pattern" lib_A.so lib_B.so ...
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
PAT=$1; shift
for src in "$@"; do
    export PPO_HIP_LIBRARY="$(realpath "$src")"
    D=/tmp/c4_kab_$(basename $src)
    rm -rf $D
    rocprofv3 --kernel-trace --stats -f csv -d $D -o run -- python3 tools/config4_bench.py > $D.out 2>&1
    python3 - "$D" "$PAT" "$src" <<'PY'
import csv, glob, re, sys
d, pat, src = sys.argv[1:4]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
if not f:
    print(src, "NO TRACE"); sys.exit(0)
out = []
for r in csv.DictReader(open(f[0])):
    if re.search(pat, r["Name"]):
        out.append("%s %s x %.1f us" % (re.sub(r"\(anonymous namespace\)::|\(.*|void ", "", r["Name"])[:34], r["Calls"], float(r["AverageNs"]) / 1e3))
print("%-34s" % src, "; ".join(out))
PY
done
