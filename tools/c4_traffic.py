#!/usr/bin/env python3
"""HBM traffic of ONE minibatch step of the generic (configs[4]) path from a --pmc summary and the kernel trace of the same command:
   python tools/c4_traffic.py PMC_PER_DISPATCH.json KERNEL_STATS.csv  ->  JSON on stdout
Per kernel: bytes per dispatch (FETCH_SIZE x 2 for these coalesced streams + WRITE_SIZE, tools/pmc_summary.py) x dispatches per optimizer step
(dispatch count of the trace / 280 optimizer steps of tools/config4_bench.py's 7 iterations); kernels that run once per iteration (rollout, GAE, ...)
are listed apart and are not part of the step -- except the update's own once-per-update work (the observations rounded to bf16 for the steps' in-place reads),
whose bytes are spread over the update's 40 steps."""
import csv
import json
import re
import sys

pmc = json.load(open(sys.argv[1]))
calls = {}
for row in csv.DictReader(open(sys.argv[2])):
    n = re.sub(r"\(anonymous namespace\)::", "", row["Name"])
    n = re.sub(r"^void\s+", "", n)
    depth, out = 0, []
    for ch in n:
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    calls["".join(out).strip()] = int(row["Calls"])
STEPS = 280
AMORTIZED = ("to_bf16_pad_kernel", "pack_rows_kernel")   # once per update, for the update's steps
step, per_iter = {}, {}
for k, v in pmc.items():
    if "hbm_read_bytes" not in v or k not in calls:
        continue
    per_step = calls[k] / STEPS
    b = v["hbm_read_bytes"] + v["hbm_write_bytes"]
    (step if per_step >= 0.9 or k.startswith(AMORTIZED) else per_iter)[k] = {"bytes_per_dispatch": b, "read": v["hbm_read_bytes"], "write": v["hbm_write_bytes"], "dispatches_per_step": per_step,
                                                  "bytes_per_step": b * per_step}
tot = sum(x["bytes_per_step"] for x in step.values())
print(json.dumps({"bytes_per_minibatch_step": tot, "read_per_step": sum(x["read"] * x["dispatches_per_step"] for x in step.values()),
                  "write_per_step": sum(x["write"] * x["dispatches_per_step"] for x in step.values()), "kernels_of_a_step": step, "once_per_iteration": per_iter}, indent=1))
