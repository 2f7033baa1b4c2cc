#!/usr/bin/env python3
"""Diagnostic: where one wave of the dominant kernel (fused forward/backward, MFMA) spends its cycles.

Runs the stamped kernel variant (ppo_profile_enable(ctx, 3)): s_memtime at every phase boundary of wave 0 / workgroup 0 of
each net.  Read the SHARES, not the run time (the stamps' fences forbid overlaps the shipped kernel has).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

P = load_package()
ctx = P.Context(P.make_config(num_envs=4096, num_steps=128, num_minibatches=4, update_epochs=10, seed=2, total_timesteps=4096 * 128 * 10))
ctx.init_orthogonal(2)
ctx.env_reset()
for _ in range(2):
    ctx.train_iteration()
ctx.profile_enable(3)
ctx.train_iteration()
p = ctx.profile_read()
names = ["prologue", "gather", "L1+tanh", "L2mfma+tanh", "head+loss", "h2img+dW3", "dz2+imgs+opA", "dW2mfma", "dh1mfma+dz1", "dz1img+dW1",
         "epilogue", "partner_wave_loop"]
for net in (0, 1):
    ph = p["phase_cycles"][net * 12:net * 12 + 12]
    tot = sum(ph)
    print("net", net, "cycles per launch %.0f" % (tot / 40), " ".join("%s=%.1f%%" % (n, 100 * v / tot) for n, v in zip(names, ph) if v))
ctx.profile_enable(1)
for _ in range(3):
    ctx.train_iteration()
p = ctx.profile_read()
print({k: round(1e3 * p[k + "_ms"] / max(p[k + "_launches"], 1), 2) for k in ("fwd_bwd", "gae", "rollout", "optimizer", "reduce")}, "us per launch")
ctx.close()
