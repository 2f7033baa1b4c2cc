#!/usr/bin/env python3
"""Throughput of BASELINE configs[3] (MountainCar, 8192 envs x 128 steps, CategoricalMasked path) on one GPU: env-steps/s of whole
iterations (rollout + GAE + update), same timing discipline as bench.py.  Not the headline workload; a guard against a slow path."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
P = load_package()
N, T, K, W = 8192, 128, 20, 3
ctx = P.Context(P.make_config(env_kind=P.ENV_MOUNTAINCAR, dist_kind=P.DIST_MASKED, obs_size=2, head_dims=(3,), num_envs=N, num_steps=T,
                              num_minibatches=4, update_epochs=10, max_episode_steps=200, seed=1, total_timesteps=(K + W) * N * T,
                              learning_rate=1e-3, gamma=0.99, gae_lambda=0.95, ent_coef=0.01))
ctx.init_orthogonal(1); ctx.env_reset()
for _ in range(W):
    ctx.train_iteration()
ctx.profile_enable(1); ctx.sync()
t0 = time.perf_counter()
for _ in range(K):
    ctx.train_iteration()
ctx.sync()
dt = time.perf_counter() - t0
p = ctx.profile_read(); st = ctx.stats()
print(json.dumps({"workload": "MountainCar masked, %d envs x %d steps" % (N, T), "env_steps_per_s": K * N * T / dt, "ms_per_iteration": 1e3 * dt / K,
                  "us_per_update_launch": 1e3 * p["fwd_bwd_ms"] / max(p["fwd_bwd_launches"], 1), "rollout_ms": p["rollout_ms"] / K,
                  "entropy": st["entropy_loss"], "loss": st["loss"]}))
ctx.close()
