#!/usr/bin/env python3
"""Times the critic of configs[4]'s network over ROWS rows (one fused launch, or the chain of layer products, whichever the library picks)."""
import ctypes as C
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
P = load_package()
rows = int(os.environ.get("ROWS", 65536))
ctx = P.Context(P.make_config(env_kind=P.ENV_SYNTHETIC, dist_kind=P.DIST_MASKED, obs_size=376, head_dims=(3, 3, 3, 2), hidden=256, n_hidden=4, num_envs=2048,
                              num_steps=128, num_minibatches=4, update_epochs=10, seed=1, total_timesteps=1 << 24, compute_dtype=P.DTYPE_BF16))
ctx.init_orthogonal(1)
obs = ctx.dev(np.random.default_rng(0).standard_normal((rows, 376)).astype(np.float32))
out = ctx.empty((rows + 64,), np.float32)
lib = P.binding.lib()
for _ in range(3):
    lib.ppo_get_value(ctx.h, obs.ptr, C.c_int64(rows), out.ptr)
ctx.sync()
t0 = time.perf_counter()
n = int(os.environ.get("REPS", 20))
for _ in range(n):
    lib.ppo_get_value(ctx.h, obs.ptr, C.c_int64(rows), out.ptr)
ctx.sync()
print("rows", rows, "us per critic pass", round(1e6 * (time.perf_counter() - t0) / n, 1))
if os.environ.get("STAMPS"):   # diagnostic build (-DFU_DBG_STAMPS): cycle stamps of wave 0 / workgroup 0 on its second tile, in the last 64 outputs
    st = out.download().view(np.uint32)[-64:].astype(np.int64)
    names = {0: "top", 1: "staged", 2: "fetch issued", 3: "barrier", 30: "out stored"}
    for l in range(5):
        names[8 + 2 * l] = "layer %d done" % l
        names[9 + 2 * l] = "layer %d barrier" % l
    sub = {}
    for l in range(5):
        sub[32 + 4 * l] = "L%d bias issued" % l; sub[33 + 4 * l] = "L%d products done" % l; sub[34 + 4 * l] = "L%d next frags issued" % l
        if os.environ.get("STAMPS") == "2":   # a scratch build with two more stamps per layer (top of the layer; its scalars have arrived)
            sub[52 + 2 * l] = "L%d top" % l; sub[53 + 2 * l] = "L%d scalars in" % l
    prev = st[0]
    for i in sorted(names):
        print("  %-16s +%6d cycles (%6d since top)" % (names[i], (st[i] - prev) & 0xffffffff, (st[i] - st[0]) & 0xffffffff))
        prev = st[i]
    for i in sorted(sub, key=lambda i: (st[i] - st[0]) & 0xffffffff):
        print("  %-22s %6d since top" % (sub[i], (st[i] - st[0]) & 0xffffffff))
