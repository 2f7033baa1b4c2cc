#!/usr/bin/env python3
"""Diagnostic: the MEASURED distance between the HIP path and the reference's fixtures for the quantities tests/test_gpu_parity.py bounds
(gradients, AdamW moments, parameters of the first two optimizer steps; the float total norm of the injected-gradient AdamW test), so the
tolerances in the tests can be set at ~3x what is measured instead of by feel.  Prints one JSON object per fixture."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402
import test_gpu_parity as TP  # noqa: E402

P = load_package()
for name in TP.DISCRETE + TP.MASKED:
    g, meta = TP.load(name)
    U = "u1/"
    B = meta["T"] * meta["N"]
    MB = B // meta["nmb"]
    ctx = TP.make_ctx(P, meta)
    TP._load_batch(ctx, g, U, meta)
    ctx.set_params(g[U + "params_before"])
    ctx.set_learning_rate(float(g[U + "lr"][0]))
    out = dict(fixture=name, steps=[])
    for k in (0, 1):
        K = U + "k%d/" % k
        grads = ctx.minibatch_forward_backward(g[U + "perms"][0, k * MB:(k + 1) * MB])
        ctx.optimizer_step()
        m, v, _ = ctx.get_optimizer()
        rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
        out["steps"].append(dict(grad_of_max=rel(grads, g[K + "grads"]), exp_avg_of_max=rel(m, g[K + "exp_avg"]), exp_avg_sq_of_max=rel(v, g[K + "exp_avg_sq"]),
                                 params_abs=float(np.abs(ctx.get_params() - g[K + "params_after"]).max())))
    ctx.close()
    if not meta["masked"]:
        ctx = TP.make_ctx(P, meta)
        ctx.set_params(g[U + "params_before"])
        ctx.set_learning_rate(float(g[U + "lr"][0]))
        ctx.write("GRADS", g[U + "k0/grads"])
        ctx.optimizer_step()
        tn = ctx.stats()["total_norm"]
        out["adamw_total_norm"] = dict(device=float(np.float32(tn)), reference=float(np.float32(g[U + "step_scalars"][0, 6])),
                                       same_float=bool(np.float32(tn) == np.float32(g[U + "step_scalars"][0, 6])))
        ctx.close()
    print(json.dumps(out), flush=True)
