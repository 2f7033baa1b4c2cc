#!/usr/bin/env python3
"""Soak: many training iterations of the bench workload (and shorter runs of configs[4]'s network in f32 and in bf16 storage), checking after every block that the
statistics stay finite and that CartPole is solved and STAYS solved -- a cheap net for rare races (no sanitizer exists for the GPU side).
  python tools/soak.py [iterations]"""
import json, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402
P = load_package()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
N, T = 4096, 128
ctx = P.Context(P.make_config(num_envs=N, num_steps=T, num_minibatches=4, update_epochs=10, max_episode_steps=500, seed=2, total_timesteps=iters * N * T,
                              learning_rate=1e-3, gamma=0.98, gae_lambda=0.95, anneal_lr=True))
ctx.init_orthogonal(2); ctx.env_reset()
t0 = time.perf_counter(); best = 0.0
for i in range(iters):
    ctx.train_iteration()
    if (i + 1) % 100 == 0:
        st = ctx.stats()
        assert all(math.isfinite(st[k]) for k in ("loss", "pg_loss", "v_loss", "approx_kl", "total_norm", "ep_len_mean")), st
        best = max(best, st["ep_len_mean"])
        print(json.dumps({"iteration": i + 1, "ep_len_mean": round(st["ep_len_mean"], 1), "loss": round(st["loss"], 4), "kl": round(st["approx_kl"], 5),
                          "elapsed_s": round(time.perf_counter() - t0, 1)}), flush=True)
assert best >= 475.0, best
ctx.close()
g = P.Context(P.make_config(env_kind=P.ENV_SYNTHETIC, dist_kind=P.DIST_MASKED, obs_size=376, head_dims=(3, 3, 3, 2), hidden=256, n_hidden=4, num_envs=2048,
                            num_steps=128, num_minibatches=4, update_epochs=10, max_episode_steps=200, seed=1, total_timesteps=40 * 2048 * 128))
g.init_orthogonal(1); g.env_reset()
for i in range(40):
    g.train_iteration()
st = g.stats()
assert all(math.isfinite(st[k]) for k in ("loss", "pg_loss", "v_loss", "approx_kl", "total_norm")), st
print(json.dumps({"generic_iterations": 40, "loss": st["loss"], "entropy": st["entropy_loss"]}))
g.close()
# the same network with bf16 storage: the two nets' passes run on two streams -- 60 iterations twice, the parameters must agree bit for bit
import numpy as np
finals = []
for rep in range(2):
    b = P.Context(P.make_config(env_kind=P.ENV_SYNTHETIC, dist_kind=P.DIST_MASKED, obs_size=376, head_dims=(3, 3, 3, 2), hidden=256, n_hidden=4, num_envs=2048,
                                num_steps=128, num_minibatches=4, update_epochs=10, max_episode_steps=200, seed=1, total_timesteps=60 * 2048 * 128,
                                compute_dtype=P.DTYPE_BF16))
    b.init_orthogonal(1); b.env_reset()
    for i in range(60):
        b.train_iteration()
    st = b.stats()
    assert all(math.isfinite(st[k]) for k in ("loss", "pg_loss", "v_loss", "approx_kl", "total_norm")), st
    finals.append(b.get_params().view(np.uint32).copy())
    b.close()
assert np.array_equal(finals[0], finals[1]), int((finals[0] != finals[1]).sum())
print(json.dumps({"bf16_iterations": 60, "runs": 2, "bit_identical": True, "loss": st["loss"]}))
print("soak ok")
