#!/usr/bin/env python3
"""Throughput of one GPU's share of BASELINE configs[4] (synthetic env, obs 376, heads [3,3,3,2], 4 x 256 MLP; 16 384 envs over 8 GPUs =
2048 envs x 128 steps per GPU, 4 minibatches x 10 epochs): env-steps/s of whole iterations, same timing discipline as bench.py.
Layers: kernels_gemm.hip.  DTYPE=bf16 (default; configs[4]'s own arithmetic: bf16 operands and stored activations, f32 accumulation) or DTYPE=f32
(f32 carried as three bf16 terms)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
P = load_package()
N, T, K, W = int(os.environ.get("ENVS", 2048)), 128, 5, 2
ctx = P.Context(P.make_config(env_kind=P.ENV_SYNTHETIC, dist_kind=P.DIST_MASKED, obs_size=376, head_dims=(3, 3, 3, 2), hidden=256, n_hidden=4, num_envs=N,
                              num_steps=T, num_minibatches=4, update_epochs=10, max_episode_steps=200, seed=1, total_timesteps=(K + W) * N * T,
                              learning_rate=3e-4, gamma=0.99, gae_lambda=0.95, ent_coef=0.01,
                              compute_dtype=P.DTYPE_F32 if os.environ.get("DTYPE", "bf16") == "f32" else P.DTYPE_BF16, kernel_flags=int(os.environ.get("KFLAGS", 0))))
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
M = N * T // 4
flops = 3 * 2 * 2 * (376 * 256 + 3 * 256 * 256) * M + 3 * 2 * (256 * 1 + 256 * 11) * M   # forward + 2x backward, both nets
fb_ms = p["fwd_bwd_ms"] / max(p["fwd_bwd_launches"], 1)
print(json.dumps({"dtype": os.environ.get("DTYPE", "bf16"), "workload": "configs[4] per GPU: synthetic env, obs 376, 4x256, heads [3,3,3,2], %d envs x %d steps" % (N, T),
                  "env_steps_per_s": K * N * T / dt, "ms_per_iteration": 1e3 * dt / K, "rollout_ms": p["rollout_ms"] / K,
                  # minibatch_step_ms: HIP events around gather + forward + loss + backward of a step (with bf16 storage the next step's gather runs ahead,
                  # beside the optimizer tail, so only an update's first gather is inside); update_ms_per_step: everything between the rollout and the
                  # next one, per optimizer step -- (iteration - rollout) / 40 by the wall clock
                  "minibatch_step_ms": fb_ms, "minibatch_step_TFLOPs": flops / (fb_ms * 1e-3) / 1e12, "optimizer_ms": p["optimizer_ms"] / max(p["optimizer_launches"], 1),
                  "update_ms_per_step": (1e3 * dt / K - p["rollout_ms"] / K) / 40,
                  "loss": st["loss"], "entropy": st["entropy_loss"]}))
ctx.close()
