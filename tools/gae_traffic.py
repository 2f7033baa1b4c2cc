#!/usr/bin/env python3
"""Memory-side traffic of the GAE scan by size: folds rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate runs of tools/gae_sweep.py) into
one JSON, per (kernel, env count) the mean per dispatch, corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE in KiB, wide coalesced
reads tallied at half their size: x2 -- confirmed on this pool by tools/probes/fetch_calib.hip; WRITE_SIZE as read), next to the algorithmic bytes.

    tools/gae_traffic.py OUT.json PASS_DIR [PASS_DIR ...] [T=128]
"""
import csv
import glob
import json
import os
import sys

out_path, dirs = sys.argv[1], [a for a in sys.argv[2:] if not a.isdigit()]
T = next((int(a) for a in sys.argv[2:] if a.isdigit()), 128)
acc = {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            name = row["Kernel_Name"]
            tag = "gae_pipe_kernel" if "gae_pipe_kernel" in name else ("gae_kernel" if "gae_kernel" in name else None)
            if tag is None:
                continue
            short = name[name.index(tag):].split("(")[0]
            epb = int(short.split("<")[1].split(",")[0])
            envs = int(row["Grid_Size"]) // int(row["Workgroup_Size"]) * epb
            s = acc.setdefault((short, envs), {}).setdefault(row["Counter_Name"], [0.0, 0])
            s[0] += float(row["Counter_Value"])
            s[1] += 1
rows = []
for (short, envs), c in sorted(acc.items(), key=lambda kv: kv[0][1]):
    rd = c.get("FETCH_SIZE", [0.0, 1])
    wr = c.get("WRITE_SIZE", [0.0, 1])
    read_b, write_b = rd[0] / rd[1] * 1024.0 * 2.0, wr[0] / wr[1] * 1024.0
    alg_r, alg_w = 12 * envs * T + 8 * envs, 8 * envs * T
    rows.append({"kernel": short, "envs": envs, "dispatches": rd[1], "hbm_read_bytes": read_b, "hbm_write_bytes": write_b, "algorithmic_read_bytes": alg_r,
                 "algorithmic_write_bytes": alg_w, "read_ratio": read_b / alg_r, "write_ratio": write_b / alg_w if write_b else None,
                 "total_ratio": (read_b + write_b) / (alg_r + alg_w)})
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/gae_sweep.py; FETCH_SIZE x 1024 x 2 (gfx950 correction), WRITE_SIZE x 1024",
           "rows": rows}, open(out_path, "w"), indent=1)
for r in rows:
    print(r["envs"], "read %.2f MB (%.2fx)" % (r["hbm_read_bytes"] / 1e6, r["read_ratio"]), "write %.2f MB" % (r["hbm_write_bytes"] / 1e6), "total %.2fx" % r["total_ratio"])
