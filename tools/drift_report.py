#!/usr/bin/env python3
"""Diagnostic: how far do the HIP path and the CPU oracle drift from the reference's recorded loss scalars over the 40 optimizer steps
of update 1 (same batch, same permutations, same start)?  Prints, per fixture, the largest relative deviation per stat in windows of
steps, for HIP-vs-reference, oracle-vs-reference and HIP-vs-oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402
import test_gpu_parity as TP  # noqa: E402

P = load_package()
KEYS = ("pg_loss", "v_loss", "entropy_loss", "approx_kl", "clipfrac", "loss")
for name in TP.DISCRETE:
    g, meta = TP.load(name)
    U = "u1/"
    B = meta["T"] * meta["N"]
    MB = B // meta["nmb"]
    ctx = TP.make_ctx(P, meta)
    TP._load_batch(ctx, g, U, meta)
    ctx.set_params(g[U + "params_before"])
    lr = float(g[U + "lr"][0])
    ctx.set_learning_rate(lr)
    scal = g[U + "step_scalars"]
    net = O.Net.make(meta["obs"], [meta["act"]])
    hp = O.HParams(gamma=meta["gamma"], gae_lambda=meta["lam"], clip_coef=meta["clip"], ent_coef=meta["ent"], vf_coef=meta["vf"],
                   max_grad_norm=meta["mgn"], norm_adv=meta["norm_adv"], clip_vloss=meta["clip_vloss"])
    batch = (g[U + "obs"].reshape(B, -1), g[U + "actions"].reshape(B), g[U + "logprobs"].ravel(), g[U + "gae_advantages"].ravel(),
             g[U + "gae_returns"].ravel(), g[U + "values"].ravel())
    p = g[U + "params_before"].copy()
    m, v = np.zeros_like(p), np.zeros_like(p)
    dev = np.zeros((3, scal.shape[0], len(KEYS)))
    k = 0
    for e in range(meta["epochs"]):
        for s in range(meta["nmb"]):
            idx = g[U + "perms"][e, s * MB:(s + 1) * MB]
            ctx.minibatch_forward_backward(idx)
            st = ctx.stats()
            gr, so = O.minibatch_grads(net, hp, p, *batch, idx)
            for i, key in enumerate(KEYS):
                h = st["clipfrac_last" if key == "clipfrac" else key]
                den = max(1.0, abs(scal[k, i]))
                dev[0, k, i] = abs(h - scal[k, i]) / den
                dev[1, k, i] = abs(so[key] - scal[k, i]) / den
                dev[2, k, i] = abs(h - so[key]) / den
            ctx.optimizer_step()
            gc, _ = O.clip_grad_norm(net, gr, hp.max_grad_norm)
            p, m, v = O.adamw_step(p, gc, m, v, lr, k + 1)
            k += 1
    print(name, "steps", k)
    for w0 in range(0, k, 8):
        print("  steps %2d-%2d  hip-ref %.2e  oracle-ref %.2e  hip-oracle %.2e" % (w0, min(w0 + 7, k - 1), dev[0, w0:w0 + 8].max(), dev[1, w0:w0 + 8].max(), dev[2, w0:w0 + 8].max()))
    print("  worst stat hip-ref:", KEYS[int(np.argmax(dev[0].max(0)))], " params: hip-ref %.2e oracle-ref %.2e" % (
        np.abs(ctx.get_params() - g[U + "params_after"]).max(), np.abs(p - g[U + "params_after"]).max()))
    ctx.close()
