#!/usr/bin/env python3
"""Soak of the wave-specialised update kernel's hand-over protocol (LDS event counters between forward and gradient waves; no sanitizer exists for
the GPU side): TWO contexts with identical state train side by side -- kernels of different contexts interleave differently on the chip every time
-- and after every iteration their parameters must agree BIT FOR BIT (a race in the hand-over would show as a difference; the fixed summation orders
make the result a function of the inputs alone).
  python tools/soak_ws.py [iterations] [workload: cartpole | mountaincar]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402
P = load_package()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
mc = len(sys.argv) > 2 and sys.argv[2] == "mountaincar"
kw = dict(num_steps=128, num_minibatches=4, update_epochs=10, seed=2, total_timesteps=iters * 4096 * 128, learning_rate=1e-3, gamma=0.98, gae_lambda=0.95)
if mc:
    kw.update(env_kind=P.ENV_MOUNTAINCAR, dist_kind=P.DIST_MASKED, obs_size=2, head_dims=(3,), num_envs=8192, max_episode_steps=200, ent_coef=0.01)
else:
    kw.update(num_envs=4096, max_episode_steps=500)
a, b = P.Context(P.make_config(**kw)), P.Context(P.make_config(**kw))
for x in (a, b):
    x.init_orthogonal(2); x.env_reset()
t0 = time.perf_counter()
for i in range(iters):
    a.train_iteration(); b.train_iteration()
    pa, pb = a.get_params(), b.get_params()
    assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32)), "iteration %d: the two contexts' parameters differ in %d elements" % (i, int((pa != pb).sum()))
    if (i + 1) % 50 == 0:
        st = a.stats()
        assert all(np.isfinite(st[k]) for k in ("loss", "pg_loss", "v_loss", "approx_kl", "total_norm")), st
        print(json.dumps({"iteration": i + 1, "ep_len_mean": round(st["ep_len_mean"], 1), "loss": round(st["loss"], 4), "elapsed_s": round(time.perf_counter() - t0, 1)}), flush=True)
print(json.dumps({"iterations": iters, "workload": "mountaincar" if mc else "cartpole", "bit_identical_every_iteration": True}))
for x in (a, b):
    x.close()
