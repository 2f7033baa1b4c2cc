#!/bin/bash
# In-trace (rocprofv3 --kernel-trace --stats) average durations of named kernels for several builds of libppo_hip.so inside ONE gpurun call, alternated:
#   tools/trace_ab.sh ROUNDS "kernel_a|kernel_b" lib_A.so lib_B.so ...        (headline workload, 20 + 3 iterations per run)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
R=$1; PAT=$2; shift; shift
for i in $(seq 1 $R); do
    for src in "$@"; do
        export PPO_HIP_LIBRARY="$(realpath "$src")"
        D=/tmp/trace_ab_${i}_$(basename $src)
        rm -rf $D
        rocprofv3 --kernel-trace --stats -f csv -d $D -o run -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads --repeats 0 > $D.out 2>&1
        python3 - "$D" "$PAT" "$src" <<'PY'
import csv, glob, re, sys
d, pat, src = sys.argv[1:4]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
if not f:
    print(src, "NO TRACE"); sys.exit(0)
out = []
for r in csv.DictReader(open(f[0])):
    if re.search(pat, r["Name"]):
        out.append("%s %s x %.2f us" % (re.sub(r"\(anonymous namespace\)::|\(.*", "", r["Name"])[:40], r["Calls"], float(r["AverageNs"]) / 1e3))
print("%-36s" % src, "; ".join(out))
PY
    done
done
