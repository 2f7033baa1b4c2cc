#!/bin/bash
# A/B of libppo_hip.so builds on the GAE scan inside ONE gpurun call: in-trace kernel durations by size, builds alternated round-robin.
#   tools/gae_ab.sh ROUNDS lib_A.so lib_B.so ...
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
R=$1; shift
for i in $(seq 1 $R); do
    for src in "$@"; do
        export PPO_HIP_LIBRARY="$(realpath "$src")"   # binding.py loads this build; the shipped library is never overwritten
        D=/tmp/gae_ab_${i}_$(basename $src)
        rm -rf $D
        rocprofv3 --kernel-trace -f csv -d $D -o run -- python3 tools/gae_sweep.py ${GAE_SIZES:-4096 8192 32768} > $D.out 2>&1
        if grep -q '"bit_exact_vs_oracle": false' $D.out || ! grep -q bit_exact $D.out; then echo "$src: WRONG RESULTS or no run"; tail -n 3 $D.out; fi
        echo "$src: $(python3 tools/gae_by_size.py "$(find $D -name '*kernel_trace.csv' | head -n 1)" /tmp/gae_ab.json | tr '\n' ';')"
    done
done
