#!/usr/bin/env python3
"""Diagnostic (a build with -DMG_STAMP): how long F wave 0 and G wave 8 of workgroup 0 of the wave-specialised update kernel sit in their waits.
Usage: PPO_HIP_LIBRARY=build_ab/libppo_hip_ws_stamp.so python tools/ws_stamps.py"""
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
for net in (0, 1):
    ph = p["phase_cycles"][net * 12:net * 12 + 12]
    f_tot, w1, w2, w3, _, g_tot, g_wait = ph[:7]
    print("net %d: F wave 0: loop %.0f ticks per launch, waits RA %.1f%% dz2->RB %.1f%% dz1->RB %.1f%% | G wave 8: loop %.0f ticks, waiting for events %.1f%%"
          % (net, f_tot / 40, 100 * w1 / f_tot, 100 * w2 / f_tot, 100 * w3 / f_tot, g_tot / 40, 100 * g_wait / max(g_tot, 1)))
    pro, whole, real = ph[7], ph[8], ph[9]
    if whole > 0 and real > 0:
        ghz = (whole - pro) / (real * 10.0)   # core ticks per ns
        print("        workgroup 0 per launch: prologue %.0f ticks, first to last instruction %.0f ticks = %.2f us at the measured %.2f GHz (F loop %.2f us, prologue %.2f us, after the F loop %.2f us)"
              % (pro / 40, whole / 40, whole / 40 / ghz / 1e3, ghz, (f_tot - pro) / 40 / ghz / 1e3, pro / 40 / ghz / 1e3, (whole - f_tot) / 40 / ghz / 1e3))
        print("        ticks from the workgroup's first instruction: F wave 0 leaves its loop %.0f, G wave 8 leaves its loop %.0f, has stored its gradient image %.0f, passes the barrier behind that %.0f, last instruction %.0f" % (f_tot / 40, g_tot / 40, ph[10] / 40, ph[11] / 40, whole / 40))
ctx.close()
