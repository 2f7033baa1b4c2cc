"""Full-size runs of BASELINE.json's single-GPU configurations through the C-ABI, checked with size-independent properties:

  configs[1]  CartPole, 4096 envs x 128 steps, 2x64 MLP, 4 minibatches (the bench workload)
  configs[3]  MountainCar, 8192 envs x 128 steps, CategoricalMasked path

At these sizes no golden fixture exists (the reference needs minutes per update on CPU), so every stage is checked against the
oracle on exactly the data the GPU produced:
  * rollout: every stored transition obs[t] --action[t]--> obs[t+1] (where no reset intervened) replays BIT-EXACT through the
    oracle's env step, rewards and done flags included; log-probs / values of a row sample agree with the oracle's forward;
    the sampled actions reproduce the oracle's Philox sampler on the same rows;
  * advantages / returns of the whole [T, N] buffer: BIT-EXACT against the oracle's scan;
  * one optimizer step on a full-size minibatch (131 072 / 262 144 rows): losses within 1e-5, gradient within 1e-4 of max,
    AdamW moments and parameters against the oracle's clip + AdamW on the GPU's own gradient.
"""
import numpy as np
import pytest

import oracle as O
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def P():
    return load_package()


def _run(P, env_kind, dist_kind, obs_dim, act, N, T, max_steps, step_fn, gamma, lam, seed):
    hp = dict(gamma=gamma, gae_lambda=lam, clip_coef=0.2, ent_coef=0.01 if dist_kind == P.DIST_MASKED else 0.0, vf_coef=0.5, max_grad_norm=0.5)
    ctx = P.Context(P.make_config(env_kind=env_kind, dist_kind=dist_kind, obs_size=obs_dim, head_dims=(act,), num_envs=N, num_steps=T,
                                  num_minibatches=4, update_epochs=2, max_episode_steps=max_steps, seed=seed, total_timesteps=4 * N * T,
                                  learning_rate=1e-3, anneal_lr=False, **hp))
    ctx.init_orthogonal(seed)
    params = ctx.get_params()
    params[-(act * 64 + act):] *= 20.0    # a policy that is not uniform, so sampling and the ratio terms are exercised
    ctx.set_params(params)
    ctx.env_reset()
    ctx.rollout()
    masked = dist_kind == P.DIST_MASKED
    net = O.Net.make(obs_dim, [act], dist_kind=O.DIST_MASKED if masked else O.DIST_CATEGORICAL)
    obs = ctx.read("OBS", (T, N, obs_dim))
    actions = ctx.read("ACTIONS", (T, N)).astype(np.int64)
    logp = ctx.read("LOGPROBS", (T, N))
    values = ctx.read("VALUES", (T, N))
    rewards = ctx.read("REWARDS", (T, N))
    dones = ctx.read("DONES", (T, N))
    next_obs = ctx.read("NEXT_OBS", (N, obs_dim))
    next_done = ctx.read("NEXT_DONE", (N,))
    next_value = ctx.read("NEXT_VALUE", (N,))
    assert actions.min() >= 0 and actions.max() == act - 1 and len(np.unique(actions)) == act

    # ---- rollout: replay every stored transition through the oracle's env step ----
    ns, r, term = step_fn(obs.reshape(T * N, obs_dim), actions.reshape(T * N))
    ns, r, term = ns.reshape(T, N, obs_dim), r.reshape(T, N), term.reshape(T, N)
    assert np.array_equal(r, rewards)                                  # reward of step t is the oracle's, always
    following = np.concatenate([obs[1:], next_obs[None]], 0)
    done_after = np.concatenate([dones[1:], next_done[None].astype(np.float32)], 0)   # m_dones[t+1] = done flag of step t
    cont = done_after == 0
    assert np.array_equal(bits(following[cont]), bits(ns[cont]))       # no reset in between: the next stored obs IS the oracle's state
    assert np.all(done_after[term != 0] == 1)                          # every termination was flagged; the rest are time-limit truncations
    assert cont.mean() > 0.5 and (~cont).sum() > 0

    # ---- policy: log-prob / value on a row sample, and the sampler itself ----
    rng = np.random.default_rng(0)
    rows = rng.choice(T * N, 8192, replace=False)
    mask = np.ones((rows.size, act), np.uint8) if masked else None
    lp_o, en_o, v_o = O.evaluate(net, params, obs.reshape(T * N, obs_dim)[rows], actions.reshape(T * N)[rows], mask)
    np.testing.assert_allclose(logp.reshape(-1)[rows], lp_o, rtol=0, atol=3e-6)
    np.testing.assert_allclose(values.reshape(-1)[rows], v_o, rtol=0, atol=3e-6)
    np.testing.assert_allclose(next_value[:512], O.get_value(net, params, next_obs[:512]), rtol=0, atol=3e-6)
    t_s = 37
    a_o, _, _, _ = O.act(net, params, obs[t_s], seed, t_s, 0, np.ones((N, act), np.uint8) if masked else None)
    agree = (a_o[:, 0] == actions[t_s]).mean()
    assert agree >= 0.999, agree                                       # identical uniforms; a pick can flip only where u sits within 1e-6 of a bin edge

    # ---- advantages / returns of the whole buffer: bit-exact ----
    ctx.calc_advantage()
    adv, ret = ctx.read("ADVANTAGES", (T, N)), ctx.read("RETURNS", (T, N))
    adv_o, ret_o = O.gae(rewards, values, dones, next_value, next_done, gamma, lam)
    assert np.array_equal(bits(adv), bits(adv_o)) and np.array_equal(bits(ret), bits(ret_o))

    # ---- one optimizer step on a full-size minibatch ----
    B = T * N
    MB = B // 4
    idx = rng.permutation(B)[:MB].astype(np.int32)
    grads = ctx.minibatch_forward_backward(idx)
    st = ctx.stats()
    hpo = O.HParams(norm_adv=1, clip_vloss=1, **hp)
    g_o, s_o = O.minibatch_grads(net, hpo, params, obs.reshape(B, obs_dim), actions.reshape(B).astype(np.float32), logp.reshape(B), adv.reshape(B),
                                 ret.reshape(B), values.reshape(B), idx.astype(np.int64),
                                 np.ones((B, act), np.uint8) if masked else None)
    for key, okey in (("pg_loss", "pg_loss"), ("v_loss", "v_loss"), ("entropy_loss", "entropy_loss"), ("approx_kl", "approx_kl"),
                      ("clipfrac_last", "clipfrac"), ("loss", "loss")):
        assert abs(st[key] - s_o[okey]) <= 1e-5 * max(1.0, abs(s_o[okey])), (key, st[key], s_o[okey])
    assert np.abs(grads - g_o).max() <= 1e-6 + 1e-4 * np.abs(g_o).max()
    ctx.set_learning_rate(1e-3)
    ctx.optimizer_step()
    m, v, step = ctx.get_optimizer()
    gc, total = O.clip_grad_norm(net, grads, 0.5)
    assert abs(ctx.stats()["total_norm"] - float(total)) <= 2e-6 * float(total)
    p_o, m_o, v_o2 = O.adamw_step(params, gc, np.zeros_like(params), np.zeros_like(params), 1e-3, 1)
    np.testing.assert_allclose(m, m_o, rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(v, v_o2, rtol=2e-5, atol=1e-18)
    np.testing.assert_allclose(ctx.get_params(), p_o, rtol=0, atol=1e-6)

    # ---- and the fused path end to end: a whole update runs and learns something finite ----
    ctx.update()
    st = ctx.stats()
    assert np.isfinite(st["loss"]) and st["optimizer_steps"] == 1 + 2 * 4
    ctx.close()


def test_config1_cartpole_4096x128(P):
    _run(P, P.ENV_CARTPOLE, P.DIST_CATEGORICAL, 4, 2, 4096, 128, 500, O.cartpole_step, 0.98, 0.95, 2)


def test_config3_mountaincar_masked_8192x128(P):
    # max_episode_steps 100 instead of the env's 200: a random policy never reaches the flag inside 128 steps, and the time-limit
    # truncation + auto-reset path should be part of the run
    _run(P, P.ENV_MOUNTAINCAR, P.DIST_MASKED, 2, 3, 8192, 128, 100, O.mountaincar_step, 0.99, 0.95, 1)


@pytest.mark.parametrize("mountaincar", [False, True])
def test_update_is_a_function_of_its_inputs_at_full_size(P, mountaincar):
    """The wave-specialised update kernel hands tiles from forward waves to gradient waves through LDS event counters (no sanitizer exists for the GPU
    side).  Two contexts with identical state train side by side at the headline size -- their kernels interleave differently on the chip every time --
    and after every iteration their parameters agree BIT FOR BIT: a race in the hand-over would show as a difference, and the fixed summation orders make
    the result a function of the inputs alone (tools/soak_ws.py is the long version: 400 iterations)."""
    kw = dict(num_steps=128, num_minibatches=4, update_epochs=10, seed=2, total_timesteps=30 * 4096 * 128, learning_rate=1e-3, gamma=0.98, gae_lambda=0.95)
    if mountaincar:
        kw.update(env_kind=P.ENV_MOUNTAINCAR, dist_kind=P.DIST_MASKED, obs_size=2, head_dims=(3,), num_envs=8192, max_episode_steps=200, ent_coef=0.01)
    else:
        kw.update(num_envs=4096, max_episode_steps=500)
    a, b = P.Context(P.make_config(**kw)), P.Context(P.make_config(**kw))
    for x in (a, b):
        x.init_orthogonal(2)
        x.env_reset()
    for i in range(12):
        a.train_iteration()
        b.train_iteration()
        pa, pb = a.get_params(), b.get_params()
        assert np.array_equal(bits(pa), bits(pb)), (i, int((pa != pb).sum()))
    assert np.isfinite(a.stats()["loss"])
    a.close()
    b.close()


@pytest.mark.parametrize("mountaincar", [False, True])
def test_update_kernels_agree_on_awkward_minibatch_sizes(P, mountaincar):
    """The wave-specialised update kernel hands tiles to waves round-robin (tile = workgroup x 8 + wave, step = workgroups x 8): minibatch sizes that leave
    partial tiles, waves without a tile, and -- beyond 1024 x 32 rows -- the two forward waves of a gradient wave with DIFFERENT tile counts (the ragged last
    round of its service loop).  Explicit minibatches of such sizes through ppo_minibatch_forward_backward on the three kernels ppo_config.kernel_flags
    selects; the vector kernel (plain fp32, pinned against the compiled reference by tests/test_gpu_parity.py) is the yardstick: losses 2e-6, gradients 5e-6
    of the largest element (measured 5e-7 at 131 072 rows, tools/headline_grad_diag.py)."""
    N, T = 1024, 64
    B = N * T
    kw = dict(num_envs=N, num_steps=T, num_minibatches=1, update_epochs=1, seed=4, total_timesteps=4 * B)
    if mountaincar:
        kw.update(env_kind=P.ENV_MOUNTAINCAR, dist_kind=P.DIST_MASKED, obs_size=2, head_dims=(3,), max_episode_steps=200, ent_coef=0.01)
    ctxs = {flags: P.Context(P.make_config(kernel_flags=flags, **kw)) for flags in (P.KERNEL_UPDATE_VECTOR, P.KERNEL_UPDATE_ONE_WAVE, 0)}
    ref = ctxs[P.KERNEL_UPDATE_VECTOR]
    ref.init_orthogonal(4)
    params = ref.get_params()
    params[-(64 * (3 if mountaincar else 2) + (3 if mountaincar else 2)):] *= 20.0    # a policy that is not uniform: ratios away from 1 after a step
    ref.set_params(params)
    ref.env_reset()
    ref.rollout()
    ref.calc_advantage()
    names = ["OBS", "ACTIONS", "LOGPROBS", "REWARDS", "DONES", "VALUES", "ADVANTAGES", "RETURNS"] + (["MASKS"] if mountaincar else [])
    O_ = 2 if mountaincar else 4
    shapes = {"OBS": (T, N, O_), "ACTIONS": (T, N, 1), "MASKS": (T, N, 3)}
    data = {n: ref.read(n, shapes.get(n, (T, N))) for n in names}
    # old log-probs that differ from the new ones (ratio != 1, some samples clipped): shift them a little
    rng = np.random.default_rng(3)
    data["LOGPROBS"] = (data["LOGPROBS"] + rng.normal(0, 0.15, (T, N))).astype(np.float32)
    for c in ctxs.values():
        c.set_params(params)
        for n in names:
            c.write(n, data[n])
    perm = rng.permutation(B).astype(np.int32)
    for M in (2, 31, 32, 33, 257, 8 * 32 + 1, 1024 * 32, 1024 * 32 + 1, 1025 * 32, 1300 * 32 + 7, B):
        idx = perm[:M]
        out = {}
        for flags, c in ctxs.items():
            g = c.minibatch_forward_backward(idx)
            out[flags] = (g, c.stats())
        g0, s0 = out[P.KERNEL_UPDATE_VECTOR]
        assert np.isfinite(g0).all() and np.abs(g0).max() > 0, M
        for flags in (P.KERNEL_UPDATE_ONE_WAVE, 0):
            g, s = out[flags]
            for key in ("pg_loss", "v_loss", "entropy_loss", "approx_kl", "loss"):
                assert abs(s[key] - s0[key]) <= 2e-6 * max(1.0, abs(s0[key])), (M, flags, key, s[key], s0[key])
            assert abs(s["clipfrac_last"] - s0["clipfrac_last"]) <= 2.0 / M + 1e-7, (M, flags)       # a count: samples on the clip threshold fall either way
            assert np.abs(g - g0).max() <= 5e-6 * np.abs(g0).max(), (M, flags, np.abs(g - g0).max() / np.abs(g0).max())
    for c in ctxs.values():
        c.close()
