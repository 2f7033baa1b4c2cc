#!/usr/bin/env python3
"""Diagnostic: wall time per ppo_train_iteration at BASELINE configs[1] (100 iterations, no per-kernel events) -- the number an A/B of two builds of
libppo_hip.so compares when the difference is in the once-per-update kernels."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
P = load_package()
ctx = P.Context(P.make_config(num_envs=4096, num_steps=128, num_minibatches=4, update_epochs=10, seed=2, total_timesteps=4096 * 128 * 50))
ctx.init_orthogonal(2); ctx.env_reset()
for _ in range(3): ctx.train_iteration()
ctx.sync(); t0 = time.perf_counter()
for _ in range(100): ctx.train_iteration()
ctx.sync(); print("ms/iter %.4f" % (1e3 * (time.perf_counter() - t0) / 100))
ctx.close()
