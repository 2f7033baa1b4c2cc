#!/usr/bin/env python3
"""Gradient of one minibatch step at configs[4]'s shape (bf16) for several minibatch sizes, saved for comparison between library builds:
   PPO_HIP_LIBRARY=build_ab/libppo_hip_X.so python tools/bwd_check.py OUT.npz   then   python tools/bwd_check.py --compare A.npz B.npz"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    shapes = [(256, 376), (256, 1), (256, 256), (256, 1), (256, 256), (256, 1), (256, 256), (256, 1), (1, 256), (1, 1),
              (256, 376), (256, 1), (256, 256), (256, 1), (256, 256), (256, 1), (256, 256), (256, 1), (11, 256), (11, 1)]
    offs = np.cumsum([0] + [x * y for x, y in shapes])
    for k in a.files:
        d = np.abs(a[k] - b[k])
        per = [float(d[offs[i]:offs[i + 1]].max() / (np.abs(a[k][offs[i]:offs[i + 1]]).max() + 1e-30)) for i in range(20)]
        print(k, "max rel", d.max() / np.abs(a[k]).max(), "per tensor:", " ".join("%.0e" % x for x in per))
    sys.exit(0)
from __graft_entry__ import load_package
P = load_package()
N, T = 256, 64
ctx = P.Context(P.make_config(env_kind=P.ENV_SYNTHETIC, dist_kind=P.DIST_MASKED, obs_size=376, head_dims=(3, 3, 3, 2), hidden=256, n_hidden=4, num_envs=N, num_steps=T,
                              num_minibatches=1, update_epochs=1, max_episode_steps=50, seed=3, total_timesteps=4 * N * T, ent_coef=0.01, compute_dtype=P.DTYPE_BF16))
ctx.init_orthogonal(3)
ctx.env_reset(); ctx.rollout(); ctx.calc_advantage()
out = {}
rng = np.random.default_rng(0)
for M in (64, 100, 128, 192, 1024, 2048, 4096, 16384):
    idx = rng.permutation(N * T)[:M].astype(np.int32)
    out["M%d" % M] = ctx.minibatch_forward_backward(idx)
np.savez(sys.argv[1], **out)
ctx.close()
