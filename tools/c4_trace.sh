#!/bin/bash
# configs[4] share: generic-path tests (optional), the bench line, and the in-trace kernel durations, in one gpurun call:  tools/c4_trace.sh [notest]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
if [ "$1" != "notest" ]; then timeout -k 10 500 python -m pytest tests/test_gpu_generic.py -q -x --timeout 300 2>&1 | tail -3; fi
timeout -k 10 200 python tools/config4_bench.py 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k: round(d[k],4) for k in ('env_steps_per_s','minibatch_step_ms','update_ms_per_step','rollout_ms','loss')})"
rm -rf gpurun_out/c4trace
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/c4trace -o run -- python3 tools/config4_bench.py > gpurun_out/c4trace.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/c4trace/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:9]:
    print(r["Name"][:80].replace("(anonymous namespace)::",""), r["Calls"], round(float(r["AverageNs"])/1e3,1), round(100*float(r["TotalDurationNs"])/tot,1))
PY
