#!/usr/bin/env python3
"""Diagnostic: deviations of the bf16-storage HIP path from the oracle's bf16 mode at BASELINE configs[4]'s network shape (what the
tolerances of tests/test_gpu_generic.py::test_config4_shape_in_bf16 are set from), and of both from the f32 arithmetic."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402
P = load_package()
obs_dim, hidden, n_hidden, heads, N, T, nmb, seed = 376, 256, 4, (3, 3, 3, 2), 256, 32, 4, 3
A, H = sum(heads), len(heads)
hp = dict(gamma=0.99, gae_lambda=0.95, clip_coef=0.2, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5)
res = {}
for dtype in (1, 0):
    ctx = P.Context(P.make_config(env_kind=P.ENV_SYNTHETIC, dist_kind=P.DIST_MASKED, obs_size=obs_dim, head_dims=heads, hidden=hidden, n_hidden=n_hidden,
                                  num_envs=N, num_steps=T, num_minibatches=nmb, update_epochs=2, max_episode_steps=25, seed=seed,
                                  total_timesteps=8 * N * T, learning_rate=1e-3, anneal_lr=False, compute_dtype=dtype, **hp))
    net = O.Net.make(obs_dim, list(heads), hidden=hidden, n_hidden=n_hidden, dist_kind=O.DIST_MASKED, dtype=dtype)
    ctx.init_orthogonal(seed)
    params = ctx.get_params()
    params[-(A * hidden + A):] *= 30.0
    ctx.set_params(params)
    ctx.env_reset(); ctx.rollout()
    obs = ctx.read("OBS", (T * N, obs_dim)); masks = ctx.read("MASKS", (T * N, A)); actions = ctx.read("ACTIONS", (T * N, H)).astype(np.int64)
    logp, values = ctx.read("LOGPROBS", (T * N,)), ctx.read("VALUES", (T * N,))
    rows = np.random.default_rng(0).choice(T * N, 2048, replace=False)
    lp_o, en_o, v_o = O.evaluate(net, params, obs[rows], actions[rows], masks[rows])
    adv, ret = ctx.calc_advantage()
    B = T * N
    idx = np.random.default_rng(1).permutation(B)[:B // nmb].astype(np.int32)
    grads = ctx.minibatch_forward_backward(idx)
    st = ctx.stats()
    g_o, s_o = O.minibatch_grads(net, O.HParams(norm_adv=1, clip_vloss=1, **hp), params, obs, actions.astype(np.float32), logp, adv.reshape(B), ret.reshape(B), values,
                                 idx.astype(np.int64), masks)[:2]
    dl = max(abs(st[k] - s_o[o]) / max(1.0, abs(s_o[o])) for k, o in (("pg_loss", "pg_loss"), ("v_loss", "v_loss"), ("entropy_loss", "entropy_loss"), ("approx_kl", "approx_kl"), ("loss", "loss")))
    print("dtype %s: logprob max|d| %.2e (mean %.2e)  value max|d| %.2e (mean %.2e)  losses rel %.2e  clipfrac %g/%g  grad max|d|/max|g| %.2e" % (
        "bf16" if dtype else "f32", np.abs(logp[rows] - lp_o).max(), np.abs(logp[rows] - lp_o).mean(), np.abs(values[rows] - v_o).max(), np.abs(values[rows] - v_o).mean(),
        dl, st["clipfrac_last"], s_o["clipfrac"], np.abs(grads - g_o).max() / np.abs(g_o).max()))
    res[dtype] = (lp_o, v_o, g_o)
    ctx.close()
print("oracle bf16 vs oracle f32 (what the reduced precision itself costs): logprob %.2e  value %.2e  grad %.2e of max" % (
    np.abs(res[1][0] - res[0][0]).max(), np.abs(res[1][1] - res[0][1]).max(), np.abs(res[1][2] - res[0][2]).max() / np.abs(res[0][2]).max()))
