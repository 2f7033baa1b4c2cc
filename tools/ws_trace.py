#!/usr/bin/env python3
"""Diagnostic (a build with -DMG_STAMP -DMG_TRACE): the hand-over timeline of the third tile round of the actor's workgroup 0 in the wave-specialised
update kernel -- F wave 0 ("a" of G wave 8), F wave 5 (its "b"), G wave 8 -- as average shader cycles since the workgroup's first instruction.
Usage: PPO_HIP_LIBRARY=build_ab/libppo_hip_ws_trace.so python tools/ws_trace.py"""
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
t = [v / 40 for v in ctx.profile_read()["phase_cycles"]]
ctx.close()
names = ["Fa tile start", "Fa E0 posted", "Fa E1: wait for G begins", "Fa E1: wait ends", "Fa E1 posted", "Fa E2: wait for G begins", "Fa E2: wait ends", "Fa E2 posted",
         "Fb tile start", "Fb E0 posted", "Fb E1 posted", "Fb E2 posted",
         "G  E0a: wait begins", "G  E0a: wait ends", "G  E0a served", "G  E0b: wait ends", "G  E0b served", "G  E1a: wait ends", "G  E1a: dz2 read, region released", "G  E1a served",
         "G  E1b: wait ends", "G  E1b: region released", "G  E1b served", "G  E2b served (round ends)"]
t0 = t[0]
for k in sorted(range(24), key=lambda i: t[i]):
    print("%8.0f  %s" % (t[k] - t0, names[k]))
