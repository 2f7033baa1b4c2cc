#!/usr/bin/env python3
"""Diagnostic: rollout kernel time with sampling (Philox + categorical draw) and teacher-forced (actions read from memory)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
P = load_package()
ctx = P.Context(P.make_config(num_envs=4096, num_steps=128, num_minibatches=4, update_epochs=1, seed=2, total_timesteps=4096 * 128 * 100))
ctx.init_orthogonal(2); ctx.env_reset()
for forced in (False, True, False, True):
    acts = None
    if forced:
        acts = ctx.read("ACTIONS", (128, 4096)).astype(np.int64)
    ctx.rollout(acts) if forced else ctx.rollout()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.rollout(acts) if forced else ctx.rollout()
    ctx.sync()
    print("forced" if forced else "sampled", round(1e6 * (time.perf_counter() - t0) / 10, 1), "us per rollout (+values)")
ctx.close()
