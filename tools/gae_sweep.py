#!/usr/bin/env python3
"""GAE scan (exact mode) across working-set sizes, from BASELINE config 2 (10.5 MB, cache-resident) to 2^20 envs x 128 steps
(2.7 GB, beyond the 256 MB Infinity Cache).  Prints one JSON line per size: algorithmic GB/s = (20 B x T x N + 8 B x N) / time,
time = wall time of `reps` back-to-back launches / reps (the stream stays busy, so launch overhead is hidden; at the small
sizes the kernel's own latency dominates).  Results are checked against the CPU oracle on the smallest size and by a
size-independent property (a done row cuts the chain) on the largest.
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

P = load_package()
ctx = P.Context(P.make_config(num_envs=8, num_steps=4, num_minibatches=1, update_epochs=1))
T = 128
sizes = [int(a) for a in sys.argv[1:]] or [4096, 32768, 262144, 1048576]
rng = np.random.default_rng(0)
for N in sizes:
    rewards = np.where(rng.random((T, N), dtype=np.float32) < 0.05, -1.0, 1.0).astype(np.float32)
    values = rng.standard_normal((T, N), dtype=np.float32)
    dones = (rng.random((T, N), dtype=np.float32) < 0.05).astype(np.float32)
    if N > 32768:
        dones[64] = 1.0   # property check below
    nv = rng.standard_normal(N, dtype=np.float32)
    nd = (rng.random(N) < 0.05).astype(np.int32)
    d = [ctx.dev(rewards), ctx.dev(values), ctx.dev(dones), ctx.dev(nv), ctx.dev(nd)]
    adv, ret = ctx.empty((T, N), np.float32), ctx.empty((T, N), np.float32)
    for _ in range(3):
        P.gae_launch(ctx, *d, T, N, 0.98, 0.95, adv, ret)
    ctx.sync()
    reps = 200 if N <= 32768 else 20
    t0 = time.perf_counter()
    for _ in range(reps):
        P.gae_launch(ctx, *d, T, N, 0.98, 0.95, adv, ret)
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    nbytes = 20 * T * N + 8 * N
    a = adv.download()
    ok = True
    if N <= 32768:
        import oracle as O
        o_adv, o_ret = O.gae(rewards, values, dones, nv, nd, 0.98, 0.95)
        ok = bool(np.array_equal(a.view(np.uint32), o_adv.view(np.uint32)) and np.array_equal(ret.download().view(np.uint32), o_ret.view(np.uint32)))
    else:
        # rows below the all-done row 64 do not depend on anything above it: recompute rows 0..63 with the oracle on a column sample
        import oracle as O
        cols = rng.choice(N, 512, replace=False)
        o_adv, _ = O.gae(rewards[:64, cols], values[:64, cols], dones[:64, cols], values[63, cols] * 0, np.ones(512, np.int32), 0.98, 0.95)
        ok = bool(np.array_equal(a[:63, cols].view(np.uint32), o_adv[:63].view(np.uint32)))
    print(json.dumps({"kernel": "gae_kernel exact", "T": T, "N": N, "bytes": nbytes, "us": 1e6 * dt, "GBps": nbytes / dt / 1e9,
                      "frac_of_8TBps": nbytes / dt / 8e12, "bit_exact_vs_oracle": ok}), flush=True)
    for x in d + [adv, ret]:
        x.free()
ctx.close()
