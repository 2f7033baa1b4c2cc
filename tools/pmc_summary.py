#!/usr/bin/env python3
"""Folds rocprofv3 --pmc passes (<dir>/**/*_counter_collection.csv) into one JSON: per kernel, the mean of every counter per dispatch.

  python tools/pmc_summary.py OUT.json PASS_DIR [PASS_DIR ...]

Kernel names are reduced to the bare function name (template arguments kept).  FETCH_SIZE / WRITE_SIZE are reported by rocprofv3
in KiB; `hbm_read_bytes` applies the gfx950 correction of MI355X_MICROARCH.md "HBM" (wide coalesced reads are tallied at half
their size: x2), `hbm_write_bytes` is WRITE_SIZE as read.  That guide calls other access widths uncalibrated, so the one kernel of this
build whose reads are a random gather of 32-byte records (fwd_bwd_mfma_kernel) is calibrated with tools/probes/fetch_calib.hip: a
coalesced 64 MiB stream reports exactly half (x2 confirmed), the gather reports 15.8 MB for 262 144 records that occupy 16.8 MB of
64-byte sectors (8.4 MB useful) -- one 64-byte request per record, counted at its size -- so gather kernels get x1, and both
readings are kept in the JSON (`hbm_read_bytes_x1`, `hbm_read_bytes_x2`).
"""
GATHER_KERNELS = ("fwd_bwd_mfma_kernel", "fwd_bwd_mfma_ws_kernel", "gather_read")
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void\s+", "", name)
    depth, out = 0, []
    for ch in name:           # drop the argument list, keep template arguments
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()


def main():
    out_path, dirs = sys.argv[1], sys.argv[2:]
    acc = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    e = acc.setdefault(k, {"_res": {}, "_cnt": {}})
                    c = row["Counter_Name"]
                    s = e["_cnt"].setdefault(c, [0.0, 0])
                    s[0] += float(row["Counter_Value"])
                    s[1] += 1
                    e["_res"] = {"grid": int(row["Grid_Size"]), "workgroup": int(row["Workgroup_Size"]), "lds_bytes": int(row["LDS_Block_Size"]),
                                 "scratch_bytes": int(row["Scratch_Size"]), "vgpr": int(row["VGPR_Count"]), "agpr": int(row["Accum_VGPR_Count"]),
                                 "sgpr": int(row["SGPR_Count"])}
    res = {}
    for k, e in sorted(acc.items()):
        o = dict(e["_res"])
        for c, (tot, n) in sorted(e["_cnt"].items()):
            o[c] = tot / n
            o.setdefault("dispatches", n)
        if "FETCH_SIZE" in o:
            o["hbm_read_bytes_x1"] = o["FETCH_SIZE"] * 1024.0
            o["hbm_read_bytes_x2"] = o["FETCH_SIZE"] * 1024.0 * 2.0
            o["hbm_read_bytes"] = o["hbm_read_bytes_x1"] if k.startswith(GATHER_KERNELS) else o["hbm_read_bytes_x2"]
        if "WRITE_SIZE" in o:
            o["hbm_write_bytes"] = o["WRITE_SIZE"] * 1024.0
        res[k] = o
    with open(out_path, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    for k, o in res.items():
        print(k, {c: round(v, 1) for c, v in o.items() if c in ("dispatches", "FETCH_SIZE", "WRITE_SIZE", "hbm_read_bytes", "hbm_write_bytes",
                                                                  "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "scratch_bytes", "vgpr")})


if __name__ == "__main__":
    main()
