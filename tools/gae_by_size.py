#!/usr/bin/env python3
"""In-trace duration of every gae_kernel dispatch of a `rocprofv3 --kernel-trace` run of tools/gae_sweep.py, grouped by grid size.

    tools/gae_by_size.py <run_kernel_trace.csv> <out.json> [T=128]
"""
import csv
import json
import statistics
import sys

trace, out = sys.argv[1], sys.argv[2]
T = int(sys.argv[3]) if len(sys.argv) > 3 else 128
groups = {}
for r in csv.DictReader(open(trace)):
    name = r["Kernel_Name"]
    tag = "gae_pipe_kernel" if "gae_pipe_kernel" in name else ("gae_kernel" if "gae_kernel" in name else None)   # the scan pipelined in time / its three-phase form
    if tag is None:
        continue
    short = name[name.index(tag):].split("(")[0]
    epb = int(short.split("<")[1].split(",")[0])
    envs = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) * epb   # one workgroup per strip of epb env columns
    groups.setdefault((short, envs), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = []
for (short, envs), d in sorted(groups.items(), key=lambda kv: kv[0][1]):
    nbytes = 20 * envs * T + 8 * envs
    avg = sum(d) / len(d)
    rows.append({"kernel": short, "envs": envs, "launches": len(d), "avg_ns": avg, "min_ns": min(d), "median_ns": statistics.median(d),
                 "bytes": nbytes, "GBps": nbytes / avg, "frac_of_8TBps": nbytes / avg / 8000.0})
json.dump({"source": "rocprofv3 --kernel-trace of tools/gae_sweep.py (tools/collect_profiles.sh): in-trace duration of every gae_kernel dispatch, "
                     "grouped by grid size", "rows": rows}, open(out, "w"), indent=1)
for r in rows:
    print(r["envs"], round(r["avg_ns"]), r["median_ns"], round(r["frac_of_8TBps"], 3))
