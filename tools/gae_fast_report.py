#!/usr/bin/env python3
"""Diagnostic: how far the associative scan (ppo_gae_fast) lands from the exact walk, in units in the last place of the largest |A| the chain has
carried, and what it buys in time (back-to-back launches).  Prints one JSON object per size."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

P = load_package()
ctx = P.Context(P.make_config(num_envs=64, num_steps=8, num_minibatches=1, update_epochs=1, seed=1, total_timesteps=512))
for T, N, p_done in ((128, 4096, 0.05), (128, 4096, 0.0), (128, 8192, 0.002), (2048, 32, 0.01), (300, 1024, 0.02), (128, 32768, 0.05)):
    rng = np.random.default_rng(T * 7919 + N)
    rewards = np.where(rng.random((T, N)) < 0.05, -1.0, 1.0).astype(np.float32)
    values = rng.standard_normal((T, N)).astype(np.float32)
    dones = (rng.random((T, N)) < p_done).astype(np.float32)
    nv = rng.standard_normal(N).astype(np.float32)
    nd = (rng.random(N) < p_done).astype(np.int32)
    adv, _ = P.gae(ctx, rewards, values, dones, nv, nd, 0.98, 0.95)
    fadv, _ = P.gae(ctx, rewards, values, dones, nv, nd, 0.98, 0.95, fast=True)
    scale = np.maximum.accumulate(np.abs(adv[::-1]).astype(np.float64), axis=0)[::-1]
    ulp = np.spacing(np.maximum(scale, 1e-30).astype(np.float32)).astype(np.float64)
    err = np.abs(fadv.astype(np.float64) - adv.astype(np.float64)) / ulp
    d = [ctx.dev(rewards), ctx.dev(values), ctx.dev(dones), ctx.dev(nv), ctx.dev(nd), ctx.empty((T, N), np.float32), ctx.empty((T, N), np.float32)]
    us = {}
    for fast in (False, True):
        for _ in range(3):
            P.gae_launch(ctx, d[0], d[1], d[2], d[3], d[4], T, N, 0.98, 0.95, d[5], d[6], fast=fast)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(200):
            P.gae_launch(ctx, d[0], d[1], d[2], d[3], d[4], T, N, 0.98, 0.95, d[5], d[6], fast=fast)
        ctx.sync()
        us["fast" if fast else "exact"] = round(1e6 * (time.perf_counter() - t0) / 200, 2)
    for x in d:
        x.free()
    print(json.dumps(dict(T=T, N=N, p_done=p_done, max_ulp=float(err.max()), mean_ulp=float(err.mean()), rows_differing=float((err > 0).mean()), us=us)), flush=True)
ctx.close()
