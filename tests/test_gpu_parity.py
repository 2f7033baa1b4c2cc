"""GPU parity tests: the HIP hot path, called through the C-ABI (ppo-libtorch_amd/binding.py -> libppo_hip.so), against
(a) golden vectors produced by the compiled reference (tests/golden/*.pgld) and (b) the CPU oracle on seeded inputs.

Bars (north_star): bit-exact for env transitions, reset stream, GAE advantages/returns and the AdamW moment updates;
1e-5 (fp32) for losses; tolerances are written next to each assertion.
"""
import os

import numpy as np
import pytest

import oracle as O
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DISCRETE = ["discrete_t32_n8_seed2", "discrete_t64_n16_seed3_trunc", "discrete_t128_n64_seed1",
            # the shapes the reference ships and BASELINE.json configs[0] names, as they are (oracle/ref_harness.cpp: golden_shipped)
            "discrete_shipped_toml_t32_n8_act1", "discrete_config0_t128_n8_seed2"]
MASKED = ["multidiscrete_mountaincar_t32_n16"]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def P():
    return load_package()


def load(name):
    g = O.read_pgld(os.path.join(G, name + ".pgld"))
    m = g["meta"]
    meta = dict(T=int(m[0]), N=int(m[1]), obs=int(m[2]), act=int(m[3]), nmb=int(m[4]), epochs=int(m[5]), max_steps=int(m[6]),
                seed=int(m[7]), updates=int(m[8]), anneal=int(m[9]), use_gae=int(m[10]), norm_adv=int(m[11]), clip_vloss=int(m[12]),
                masked=int(m[13]))
    h = g["hparams"]
    meta.update(lr=float(h[0]), gamma=float(h[1]), lam=float(h[2]), clip=float(h[3]), ent=float(h[4]), vf=float(h[5]), mgn=float(h[6]))
    return g, meta


def make_ctx(P, meta, **over):
    kw = dict(env_kind=P.ENV_MOUNTAINCAR if meta["masked"] else P.ENV_CARTPOLE,
              dist_kind=P.DIST_MASKED if meta["masked"] else P.DIST_CATEGORICAL, obs_size=meta["obs"], head_dims=(meta["act"],),
              num_envs=meta["N"], num_steps=meta["T"], num_minibatches=meta["nmb"], update_epochs=meta["epochs"],
              max_episode_steps=meta["max_steps"], use_gae=True, norm_adv=bool(meta["norm_adv"]), clip_vloss=bool(meta["clip_vloss"]),
              anneal_lr=bool(meta["anneal"]), seed=meta["seed"], total_timesteps=meta["updates"] * meta["T"] * meta["N"],
              learning_rate=meta["lr"], gamma=meta["gamma"], gae_lambda=meta["lam"], clip_coef=meta["clip"], ent_coef=meta["ent"],
              vf_coef=meta["vf"], max_grad_norm=meta["mgn"])
    kw.update(over)
    return P.Context(P.make_config(**kw))


@pytest.fixture(scope="module")
def util_ctx(P):
    ctx = P.Context(P.make_config(num_envs=8, num_steps=4, num_minibatches=1, update_epochs=1))
    yield ctx
    ctx.close()


# ------------------------------------------------------------------------------------------- stateless kernels
def test_reset_stream_host_table(P):
    rs = O.read_pgld(os.path.join(G, "cartpole_reset_stream.pgld"))
    for k, v in rs.items():
        assert np.array_equal(bits(P.cartpole_reset_stream(int(k[4:]), v.shape[0])), bits(v)), k


def test_cartpole_transitions_bit_exact(P, util_ctx):
    c = O.read_pgld(os.path.join(G, "cartpole_transitions.pgld"))
    ns, r, t = P.env_transition(util_ctx, P.ENV_CARTPOLE, c["state"], c["action"])
    assert np.array_equal(bits(ns), bits(c["next_state"]))
    assert np.array_equal(r, c["reward"]) and np.array_equal(t, c["terminated"])


def test_mountaincar_transitions_bit_exact(P, util_ctx):
    c = O.read_pgld(os.path.join(G, "mountaincar_transitions.pgld"))
    ns, r, t = P.env_transition(util_ctx, P.ENV_MOUNTAINCAR, c["state"], c["action"])
    assert np.array_equal(bits(ns), bits(c["next_state"]))
    assert np.array_equal(r, c["reward"]) and np.array_equal(t, c["terminated"])


def test_transitions_empty_input(P, util_ctx):
    ns, r, t = P.env_transition(util_ctx, P.ENV_CARTPOLE, np.zeros((0, 4), np.float32), np.zeros(0, np.int64))
    assert ns.shape == (0, 4) and r.size == 0 and t.size == 0


def test_categorical_matches_reference(P, util_ctx):
    d = O.read_pgld(os.path.join(G, "distributions.pgld"))
    for name in ["cat1", "cat2", "cat3", "cat6", "masked2", "masked3", "masked6"]:
        kind = P.DIST_MASKED if name.startswith("masked") else P.DIST_CATEGORICAL
        res = P.categorical(util_ctx, kind, d[name + "/logits"], d.get(name + "/mask"), d[name + "/value"])
        for f in ["m_logits", "m_probs", "log_prob", "entropy"]:
            ref = d[name + "/" + f]
            fin = np.isfinite(ref) & (np.abs(ref) < 1e7)
            np.testing.assert_allclose(res[f][fin], ref[fin], rtol=3e-6, atol=1e-6, err_msg=name + "/" + f)  # exp/log ULPs
        assert np.array_equal(res["mode"], d[name + "/mode"])
        if kind == P.DIST_CATEGORICAL:
            assert np.all(np.abs(res["entropy"]) < 2e-38)  # the reference's clamp bug (Categorical.cpp:112-119)


@pytest.mark.parametrize("name", DISCRETE + MASKED)
def test_gae_kernel_bit_exact_vs_reference(P, util_ctx, name):
    g, meta = load(name)
    for u in range(1, meta["updates"] + 1):
        U = "u%d/" % u
        adv, ret = P.gae(util_ctx, g[U + "rewards"], g[U + "values"], g[U + "dones"], g[U + "next_value"], g[U + "next_done"],
                         meta["gamma"], meta["lam"])
        assert np.array_equal(bits(adv), bits(g[U + "gae_advantages"]))
        assert np.array_equal(bits(ret), bits(g[U + "gae_returns"]))


# The last rows drive every path of the launch selection (csrc/kernels_gae.hip, launch_scan).  gae_pipe_kernel (4096 < N <= 8192, N % 32 == 0, T % 128 == 0):
# more than one time tile (256, 384: the hand-back wait of `tile > 0`, the cumulative ready counts, the carry across tiles), no done / every step done,
# 129 strips (just above the switch), N % 128 != 0, and N % 32 != 0 (back to the three-phase kernel); both `nstep` values loop below, so the n-step
# instantiation runs on each.  The wide-strip kernel of HBM-resident sizes (N >= 65 536: 64 columns x 64-row time tiles): whole tiles, a ragged first tile
# (T = 200), and a width that is not a multiple of 64 (back to 32-column strips).
@pytest.mark.parametrize("T,N,p_done", [(128, 4096, 0.05), (128, 4096, 0.002), (300, 1024, 0.02), (5, 100, 0.3), (1, 4, 0.5),
                                        (129, 64, 0.0), (2048, 32, 0.01), (128, 32768, 0.05), (7, 1, 0.0), (130, 20, 1.0),
                                        (256, 8192, 0.05), (384, 6144, 0.002), (128, 8192, 0.0), (128, 8192, 1.0), (256, 4128, 0.3),
                                        (128, 8160, 0.05), (128, 8200, 0.05), (128, 131072, 0.05), (200, 65536, 0.01), (64, 65568, 0.05)])
def test_gae_kernel_bit_exact_vs_oracle(P, util_ctx, T, N, p_done):
    rng = np.random.default_rng(T * 1000003 + N)
    rewards = np.where(rng.random((T, N)) < 0.05, -1.0, 1.0).astype(np.float32)
    values = rng.standard_normal((T, N)).astype(np.float32)
    dones = (rng.random((T, N)) < p_done).astype(np.float32)
    nv = rng.standard_normal(N).astype(np.float32)
    nd = (rng.random(N) < p_done).astype(np.int32)
    for nstep in (False, True):
        adv, ret = P.gae(util_ctx, rewards, values, dones, nv, nd, 0.98, 0.95, nstep=nstep)
        o_adv, o_ret = (O.nstep(rewards, values, dones, nv, nd, 0.98) if nstep else O.gae(rewards, values, dones, nv, nd, 0.98, 0.95))
        assert np.array_equal(bits(adv), bits(o_adv)), (T, N, nstep)
        assert np.array_equal(bits(ret), bits(o_ret)), (T, N, nstep)


@pytest.mark.parametrize("T,N,p_done", [(128, 4096, 0.05), (128, 4096, 0.0), (300, 1024, 0.02), (5, 100, 0.3), (1, 4, 0.5), (129, 64, 0.0),
                                        (2048, 32, 0.01), (128, 32768, 0.05), (7, 1, 0.0), (130, 20, 1.0), (128, 8192, 0.002)])
def test_gae_fast_mode_stays_within_ulps(P, util_ctx, T, N, p_done):
    """ppo_gae_fast (the scan of affine maps north_star names) is NOT the reference's bit pattern -- the carry into a chunk of rows is associated
    differently -- but it stays within a few units in the last place of the quantity the chain carries: |fast - exact| <= 16 ulp(max |A| along the env's
    segment) (measured: <= 6, tools/gae_fast_report.py).  With every step a terminal one (p_done = 1) no chain is left and the two modes agree bit for bit; so do rows that close
    a chunk of the fast mode's decomposition right under a done flag."""
    rng = np.random.default_rng(T * 7919 + N)
    rewards = np.where(rng.random((T, N)) < 0.05, -1.0, 1.0).astype(np.float32)
    values = rng.standard_normal((T, N)).astype(np.float32)
    dones = (rng.random((T, N)) < p_done).astype(np.float32)
    nv = rng.standard_normal(N).astype(np.float32)
    nd = (rng.random(N) < p_done).astype(np.int32)
    adv, ret = P.gae(util_ctx, rewards, values, dones, nv, nd, 0.98, 0.95)
    f_adv, f_ret = P.gae(util_ctx, rewards, values, dones, nv, nd, 0.98, 0.95, fast=True)
    scale = np.maximum.accumulate(np.abs(adv[::-1]).astype(np.float64), axis=0)[::-1]     # the largest |A| the chain has carried on its way down to row t (crosses done flags: an upper bound)
    ulp = np.spacing(np.maximum(scale, 1e-30).astype(np.float32)).astype(np.float64)
    err = np.abs(f_adv.astype(np.float64) - adv.astype(np.float64))
    assert np.all(err <= 16 * ulp), (T, N, float((err / ulp).max()))
    assert np.array_equal(bits(f_ret), bits((f_adv + values).astype(np.float32)))            # R = A + v, one rounding, as in the exact mode
    if p_done == 1.0:
        assert np.array_equal(bits(f_adv), bits(adv))


def test_gae_segments_are_independent(P, util_ctx):
    """Size-independent property: a done at t+1 cuts the chain, so changing anything above the cut leaves rows <= t unchanged."""
    rng = np.random.default_rng(5)
    T, N = 128, 4096
    rewards = rng.standard_normal((T, N)).astype(np.float32)
    values = rng.standard_normal((T, N)).astype(np.float32)
    dones = np.zeros((T, N), np.float32)
    dones[64] = 1.0
    nv, nd = rng.standard_normal(N).astype(np.float32), np.zeros(N, np.int32)
    a1, r1 = P.gae(util_ctx, rewards, values, dones, nv, nd, 0.99, 0.95)
    rewards2, values2 = rewards.copy(), values.copy()
    rewards2[64:] += 3.0
    values2[65:] -= 1.0
    a2, r2 = P.gae(util_ctx, rewards2, values2, dones, nv + 7, nd, 0.99, 0.95)
    assert np.array_equal(bits(a1[:63]), bits(a2[:63])) and np.array_equal(bits(r1[:63]), bits(r2[:63]))
    assert not np.array_equal(bits(a1[64:]), bits(a2[64:]))


# ------------------------------------------------------------------------------------------- environments through the context
@pytest.mark.parametrize("name", DISCRETE)
def test_init_and_step_envs_bit_exact(P, name):
    """initEnvs + stepEnvs with the reference's own sampled actions: obs/reward/done bit for bit, incl. auto-reset,
    truncation at max_episode_steps, the shared reset stream and env 0's double reset."""
    g, meta = load(name)
    ctx = make_ctx(P, meta)
    obs = ctx.env_reset()
    assert np.array_equal(bits(obs), bits(g["init_obs"]))
    done = np.zeros(meta["N"], np.int32)
    U = "u1/"
    for t in range(meta["T"]):
        assert np.array_equal(bits(obs), bits(g[U + "obs"][t])), (name, t)
        assert np.array_equal(done.astype(np.float32), g[U + "dones"][t])
        obs, rew, done = ctx.env_step(g[U + "actions"][t].reshape(meta["N"], -1)[:, :1].astype(np.int64))
        assert np.array_equal(rew, g[U + "rewards"][t])
    assert np.array_equal(bits(obs), bits(g[U + "next_obs"])) and np.array_equal(done, g[U + "next_done"])
    ctx.close()


@pytest.mark.parametrize("name", DISCRETE)
def test_fused_rollout_teacher_forced(P, name):
    """The one-launch rollout with the reference's actions injected reproduces every rollout buffer of the reference:
    env-side buffers bit for bit, network outputs within fp32 noise, episode statistics exactly -- over both updates."""
    g, meta = load(name)
    ctx = make_ctx(P, meta)
    ctx.env_reset()
    T, N = meta["T"], meta["N"]
    for u in range(1, meta["updates"] + 1):
        U = "u%d/" % u
        ctx.set_params(g[U + "params_before"])
        ctx.rollout(g[U + "actions"].reshape(T, N, 1).astype(np.int64))
        assert np.array_equal(bits(ctx.read("OBS", (T, N, 4))), bits(g[U + "obs"]))
        assert np.array_equal(ctx.read("REWARDS", (T, N)), g[U + "rewards"])
        assert np.array_equal(ctx.read("DONES", (T, N)), g[U + "dones"])
        assert np.array_equal(ctx.read("ACTIONS", (T, N)), g[U + "actions"].reshape(T, N).astype(np.int32))
        assert np.array_equal(bits(ctx.read("NEXT_OBS", (N, 4))), bits(g[U + "next_obs"]))
        assert np.array_equal(ctx.read("NEXT_DONE"), g[U + "next_done"])
        np.testing.assert_allclose(ctx.read("LOGPROBS", (T, N)), g[U + "logprobs"], rtol=0, atol=3e-6)  # tanh/exp/log ULPs
        np.testing.assert_allclose(ctx.read("VALUES", (T, N)), g[U + "values"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(ctx.read("NEXT_VALUE"), g[U + "next_value"].ravel(), rtol=0, atol=3e-6)
        # calcAdvantage on the context: bit-exact GAE of the device's own values
        adv, ret = ctx.calc_advantage()
        o_adv, o_ret = O.gae(ctx.read("REWARDS", (T, N)), ctx.read("VALUES", (T, N)), ctx.read("DONES", (T, N)), ctx.read("NEXT_VALUE"),
                             ctx.read("NEXT_DONE"), meta["gamma"], meta["lam"])
        assert np.array_equal(bits(adv), bits(o_adv)) and np.array_equal(bits(ret), bits(o_ret))
        np.testing.assert_allclose(adv, g[U + "gae_advantages"], rtol=0, atol=2e-4)
        st = ctx.stats()
        ref = g[U + "ep_stats"]
        assert st["ep_count"] == int(ref[2])
        if st["ep_count"]:
            assert st["ep_len_mean"] == ref[0] and st["ep_rew_mean"] == ref[1]
    ctx.close()


def _masked_ctx_from_reference_state(P, g, meta):
    """MountainCar::reset draws from std::random_device in the reference (MountainCar.cpp:79-88): unseedable, so the reference's own initial
    observations are injected (ppo_env_set_state_h) -- from there on the env is deterministic given the actions."""
    ctx = make_ctx(P, meta)
    ctx.env_reset()
    N = meta["N"]
    ctx.env_set_state(state=g["init_obs"], ep_len=np.zeros(N, np.int32), ep_rew=np.zeros(N, np.float32))
    return ctx


@pytest.mark.parametrize("name", MASKED)
def test_masked_step_envs_bit_exact(P, name):
    """PPO_MultiDiscrete::stepEnvs (PPO_MultiDiscrete.cpp:434-504) with the reference's own sampled actions from the reference's initial state:
    obs / reward / done bit for bit against the reference's rollout trace."""
    g, meta = load(name)
    ctx = _masked_ctx_from_reference_state(P, g, meta)
    N, U = meta["N"], "u1/"
    obs, done = g["init_obs"], np.zeros(N, np.int32)
    for t in range(meta["T"]):
        assert np.array_equal(bits(obs), bits(g[U + "obs"][t])), (name, t)
        assert np.array_equal(done.astype(np.float32), g[U + "dones"][t])
        # m_actions[step] = action broadcasts [N,1] into [N,action_size] (PPO_MultiDiscrete.cpp:93,562): column 0 is the action
        obs, rew, done = ctx.env_step(g[U + "actions"][t][:, :1].astype(np.int64))
        assert np.array_equal(rew, g[U + "rewards"][t])
    assert np.array_equal(bits(obs), bits(g[U + "next_obs"])) and np.array_equal(done, g[U + "next_done"])
    ctx.close()


@pytest.mark.parametrize("name", MASKED)
def test_masked_fused_rollout_teacher_forced(P, name):
    """The MultiDiscrete twin of the rollout (PPO_MultiDiscrete.cpp:547-571: getActionMask, getActionAndValueMasked, stores, stepEnvs) as ONE launch,
    pinned to the reference's own trace: the reference's initial state and actions injected, OBS / REWARDS / DONES / MASKS / ACTIONS / NEXT_* bit for
    bit, log-prob / value / entropy-free network outputs within fp32 noise (3e-6), then calcAdvantage bit-exact on the device's own values."""
    g, meta = load(name)
    ctx = _masked_ctx_from_reference_state(P, g, meta)
    T, N, A, U = meta["T"], meta["N"], meta["act"], "u1/"
    ctx.set_params(g[U + "params_before"])
    ctx.rollout(g[U + "actions"][:, :, :1].astype(np.int64))
    assert np.array_equal(bits(ctx.read("OBS", (T, N, meta["obs"]))), bits(g[U + "obs"]))
    assert np.array_equal(ctx.read("REWARDS", (T, N)), g[U + "rewards"])
    assert np.array_equal(ctx.read("DONES", (T, N)), g[U + "dones"])
    assert np.array_equal(ctx.read("MASKS", (T, N, A)), g[U + "action_masks"].astype(np.uint8))
    assert np.array_equal(ctx.read("ACTIONS", (T, N)), g[U + "actions"][:, :, 0].astype(np.int32))
    assert np.array_equal(bits(ctx.read("NEXT_OBS", (N, meta["obs"]))), bits(g[U + "next_obs"]))
    assert np.array_equal(ctx.read("NEXT_DONE"), g[U + "next_done"])
    np.testing.assert_allclose(ctx.read("LOGPROBS", (T, N)), g[U + "logprobs"], rtol=0, atol=3e-6)  # tanh/exp/log ULPs
    np.testing.assert_allclose(ctx.read("VALUES", (T, N)), g[U + "values"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(ctx.read("NEXT_VALUE"), g[U + "next_value"].ravel(), rtol=0, atol=3e-6)
    adv, ret = ctx.calc_advantage()
    o_adv, o_ret = O.gae(ctx.read("REWARDS", (T, N)), ctx.read("VALUES", (T, N)), ctx.read("DONES", (T, N)), ctx.read("NEXT_VALUE"),
                         ctx.read("NEXT_DONE"), meta["gamma"], meta["lam"])
    assert np.array_equal(bits(adv), bits(o_adv)) and np.array_equal(bits(ret), bits(o_ret))
    np.testing.assert_allclose(adv, g[U + "gae_advantages"], rtol=0, atol=2e-4)
    st = ctx.stats()
    ref = g[U + "ep_stats"]
    assert st["ep_count"] == int(ref[2])   # T = 32 < max_episode_steps = 200 and the goal is out of reach in 32 steps: no episode ends
    ctx.close()


@pytest.mark.parametrize("name", DISCRETE + MASKED)
def test_policy_forward_teacher_forced(P, name):
    g, meta = load(name)
    ctx = make_ctx(P, meta)
    U = "u1/"
    B = meta["T"] * meta["N"]
    ctx.set_params(g[U + "params_before"])
    obs = g[U + "obs"].reshape(B, meta["obs"])
    acts = g[U + "actions"].reshape(B, -1)[:, :1].astype(np.int64)
    mask = g[U + "action_masks"].reshape(B, -1) if meta["masked"] else None
    a, lp, en, v = ctx.policy_act(obs, mask=mask, action=acts)
    assert np.array_equal(a, acts)
    np.testing.assert_allclose(lp, g[U + "logprobs"].ravel(), rtol=0, atol=3e-6)
    np.testing.assert_allclose(v, g[U + "values"].ravel(), rtol=0, atol=3e-6)
    np.testing.assert_allclose(en, g[U + "rollout_entropy"].ravel(), rtol=0, atol=3e-6)
    np.testing.assert_allclose(ctx.get_value(g[U + "next_obs"]), g[U + "next_value"].ravel(), rtol=0, atol=3e-6)
    ctx.close()


def test_sampling_matches_oracle_and_distribution(P):
    """Own counter-based sampler (the reference's torch::multinomial stream is not reproducible off LibTorch's CPU
    generator): same Philox key + inverse-CDF as the oracle, and the empirical frequencies follow softmax(logits)."""
    g, meta = load("discrete_t128_n64_seed1")
    ctx = make_ctx(P, meta)
    params = g["u1/params_before"].copy()
    # make the policy clearly non-uniform: scale the actor head
    net = O.Net.make(4, [2])
    params[-130:] *= 40.0
    ctx.set_params(params)
    rng = np.random.default_rng(0)
    obs = rng.uniform(-0.2, 0.2, (8192, 4)).astype(np.float32)
    a, lp, en, v = ctx.policy_act(obs, step_index=17)
    oa, olp, oen, ov = O.act(net, params, obs, meta["seed"], 17)
    assert (a != oa).sum() <= 2          # u within float noise of a CDF edge
    same = (a == oa).ravel()
    np.testing.assert_allclose(lp[same], olp[same], rtol=0, atol=3e-6)
    logits = O.actor_logits(net, params, obs)
    p1 = np.exp(logits[:, 1]) / np.exp(logits).sum(1)
    z = (a.ravel().sum() - p1.sum()) / np.sqrt((p1 * (1 - p1)).sum())
    assert abs(z) < 4.5, z
    assert 0.05 < p1.mean() < 0.95
    # a different step index gives a different draw
    a2, *_ = ctx.policy_act(obs, step_index=18)
    assert (a2 != a).any()
    ctx.close()


def test_multihead_masked_agent(P):
    """Agent::getActionAndValueMasked with m_actionSpace = {3,3,3,2} (the split of Agent.cpp:140-141 with more than one
    head): log-probs and entropies summed over heads, masks with disabled actions, teacher-forced actions."""
    g = O.read_pgld(os.path.join(G, "multihead_agent.pgld"))
    heads = tuple(int(h) for h in g["heads"])
    ctx = P.Context(P.make_config(dist_kind=P.DIST_MASKED, head_dims=heads, num_envs=8, num_steps=4, num_minibatches=1, update_epochs=1))
    assert ctx.P == g["params"].size
    ctx.set_params(g["params"])
    a, lp, en, v = ctx.policy_act(g["x"], mask=g["mask"], action=g["action_hn"].T)
    assert np.array_equal(a, g["action_out"])
    np.testing.assert_allclose(lp, g["logprob"], rtol=1e-5, atol=3e-6)
    np.testing.assert_allclose(en, g["entropy"], rtol=1e-5, atol=3e-6)
    np.testing.assert_allclose(v, g["value"].ravel(), rtol=1e-5, atol=3e-6)
    # sampling never picks a masked-out action
    sa, slp, *_ = ctx.policy_act(np.repeat(g["x"], 16, 0), mask=np.repeat(g["mask"], 16, 0), step_index=3)
    off = np.concatenate([[0], np.cumsum(heads)[:-1]])
    chosen = np.take_along_axis(np.repeat(g["mask"], 16, 0), sa + off[None, :], axis=1)
    assert chosen.all() and np.isfinite(slp).all()
    ctx.close()


# ------------------------------------------------------------------------------------------- update
def _load_batch(ctx, g, U, meta):
    T, N = meta["T"], meta["N"]
    ctx.write("OBS", g[U + "obs"])
    ctx.write("ACTIONS", g[U + "actions"].reshape(T, N, -1)[:, :, :1].astype(np.int32))
    ctx.write("LOGPROBS", g[U + "logprobs"])
    ctx.write("REWARDS", g[U + "rewards"])
    ctx.write("DONES", g[U + "dones"])
    ctx.write("VALUES", g[U + "values"])
    ctx.write("ADVANTAGES", g[U + "gae_advantages"])
    ctx.write("RETURNS", g[U + "gae_returns"])
    if meta["masked"]:
        ctx.write("MASKS", g[U + "action_masks"].astype(np.uint8))


@pytest.mark.parametrize("name", DISCRETE + MASKED)
def test_minibatch_step_matches_reference(P, name):
    """One optimizer step (steps 1 and 2 of update 1) on the reference's batch with the reference's minibatch indices:
    losses within 1e-5 (north_star), gradients within 5e-6 of the largest element (measured ~5e-7: 3 x the measured value would be 1.5e-6; 5e-6 is
    the bar VERDICT round 2 set), AdamW moments within 1e-5 relative of the reference's (they are bit-exact GIVEN the reference's gradient --
    test_adamw_bit_exact_given_reference_gradient; here they inherit the gradient's own 5e-7) and parameters within 1e-6."""
    g, meta = load(name)
    ctx = make_ctx(P, meta)
    U = "u1/"
    B = meta["T"] * meta["N"]
    MB = B // meta["nmb"]
    _load_batch(ctx, g, U, meta)
    ctx.set_params(g[U + "params_before"])
    ctx.set_learning_rate(float(g[U + "lr"][0]))
    scal = g[U + "step_scalars"]
    for k in (0, 1):
        K = U + "k%d/" % k
        idx = g[U + "perms"][0, k * MB:(k + 1) * MB]
        grads = ctx.minibatch_forward_backward(idx)
        st = ctx.stats()
        ref = dict(zip(O.STAT_NAMES + ("total_norm",), scal[k]))
        for n, key in (("pg_loss", "pg_loss"), ("v_loss", "v_loss"), ("entropy_loss", "entropy_loss"), ("approx_kl", "approx_kl"),
                       ("clipfrac", "clipfrac_last"), ("loss", "loss"), ("total_norm", "total_norm")):
            assert abs(st[key] - ref[n]) <= 1e-5 * max(1.0, abs(ref[n])), (name, k, n, st[key], ref[n])
        gref = g[K + "grads"]
        gmax = np.abs(gref).max()
        assert np.abs(grads - gref).max() <= 5e-6 * gmax, (name, k, np.abs(grads - gref).max() / gmax)
        ctx.optimizer_step()
        m, v, step = ctx.get_optimizer()
        assert step == k + 1
        # moments: absolute bars scaled by the largest element (an element near zero carries the ABSOLUTE error of the gradient's largest terms)
        mref, vref = g[K + "exp_avg"], g[K + "exp_avg_sq"]
        assert np.abs(m - mref).max() <= 5e-6 * np.abs(mref).max(), (name, k, np.abs(m - mref).max() / np.abs(mref).max())
        assert np.abs(v - vref).max() <= 1e-5 * np.abs(vref).max(), (name, k, np.abs(v - vref).max() / np.abs(vref).max())
        np.testing.assert_allclose(ctx.get_params(), g[K + "params_after"], rtol=0, atol=1e-6)
    ctx.close()


@pytest.mark.parametrize("name", DISCRETE)
def test_adamw_bit_exact_given_reference_gradient(P, name):
    """K10 in isolation, with no weaker branch to fall into.  clip_grad_norm_ scales the gradient by a coefficient formed from the float total norm
    (LibTorch clip_grad.h:58-85: coef = max_norm / (total + 1e-6), clamped to 1, all float); the device forms its own norm from double sums, which
    is the reference's float on four of the five fixtures and its neighbour (1 ULP) on discrete_t32_n8_seed2 (LibTorch adds the squares in float, in
    its vectorised order).  So the AdamW arithmetic is pinned on its own: the reference's gradient is clipped on the host WITH THE REFERENCE'S NORM,
    exactly as LibTorch does, and injected into a context whose max_grad_norm is out of reach (coefficient 1.0: a multiplication that changes nothing).
    Moments bit-identical on every fixture, parameters within 4 ULP (the reference's sqrt goes through MKL VML, not correctly rounded on ~0.03 % of
    elements).  The device's own norm is asserted separately: the same float where it is known to be, never further than 1 ULP."""
    g, meta = load(name)
    U, K = "u1/", "u1/k0/"
    total_ref = np.float32(g[U + "step_scalars"][0, 6])
    coef = np.float32(meta["mgn"]) / (total_ref + np.float32(1e-6))
    coef = np.float32(min(coef, np.float32(1.0)))
    clipped = (g[K + "grads"].astype(np.float32) * coef).astype(np.float32)
    ctx = make_ctx(P, meta, max_grad_norm=1e30)
    ctx.set_params(g[U + "params_before"])
    ctx.set_learning_rate(float(g[U + "lr"][0]))
    ctx.write("GRADS", clipped)
    ctx.optimizer_step()
    m, v, step = ctx.get_optimizer()
    assert step == 1
    assert np.array_equal(bits(m), bits(g[K + "exp_avg"]))
    assert np.array_equal(bits(v), bits(g[K + "exp_avg_sq"]))
    ulp = np.abs(bits(ctx.get_params()).astype(np.int64) - bits(g[K + "params_after"]).astype(np.int64))
    assert ulp.max() <= 4 and (ulp != 0).mean() <= 2e-3
    ctx.close()
    # the whole K9 + K10 step with the device's own norm
    ctx = make_ctx(P, meta)
    ctx.set_params(g[U + "params_before"])
    ctx.set_learning_rate(float(g[U + "lr"][0]))
    ctx.write("GRADS", g[K + "grads"])
    ctx.optimizer_step()
    total_dev = np.float32(ctx.stats()["total_norm"])
    ulps_apart = abs(int(bits(total_dev).item()) - int(bits(total_ref).item()))
    assert ulps_apart == (1 if name == "discrete_t32_n8_seed2" else 0), (name, float(total_dev), float(total_ref))
    m, v, _ = ctx.get_optimizer()
    if ulps_apart == 0:
        assert np.array_equal(bits(m), bits(g[K + "exp_avg"])) and np.array_equal(bits(v), bits(g[K + "exp_avg_sq"]))
    else:   # a coefficient 1 ULP off moves every clipped element by at most 1 ULP; the moments follow
        np.testing.assert_allclose(m, g[K + "exp_avg"], rtol=3e-7, atol=0)
        np.testing.assert_allclose(v, g[K + "exp_avg_sq"], rtol=6e-7, atol=0)
    ctx.close()


@pytest.mark.parametrize("name", DISCRETE)
def test_full_update_tracks_reference(P, name):
    """All epochs x minibatches of update 1 with the reference's permutations, driven step by step through the C-ABI: every step's loss
    scalars within 1e-5 of the reference's (north_star) along the WHOLE trajectory -- measured 3e-7 (tools/drift_report.py; the oracle,
    a third fp32 implementation with its own summation order, sits at the same distance) -- and the parameters after the update within 2e-6."""
    g, meta = load(name)
    ctx = make_ctx(P, meta)
    U = "u1/"
    B = meta["T"] * meta["N"]
    MB = B // meta["nmb"]
    _load_batch(ctx, g, U, meta)
    ctx.set_params(g[U + "params_before"])
    ctx.set_learning_rate(float(g[U + "lr"][0]))
    scal = g[U + "step_scalars"]
    k = 0
    for e in range(meta["epochs"]):
        for s in range(meta["nmb"]):
            ctx.minibatch_forward_backward(g[U + "perms"][e, s * MB:(s + 1) * MB])
            st = ctx.stats()
            for i, key in enumerate(("pg_loss", "v_loss", "entropy_loss", "approx_kl", "clipfrac_last", "loss")):
                assert abs(st[key] - scal[k, i]) <= 1e-5 * max(1.0, abs(scal[k, i])), (name, k, key, st[key], scal[k, i])
            ctx.optimizer_step()
            k += 1
    assert np.abs(ctx.get_params() - g[U + "params_after"]).max() <= 2e-6
    ctx.close()


def test_update_equals_stepwise_path_and_permutations_are_permutations(P):
    """ppo_update (own permutations, all minibatch statistics in one launch) == the same permutations driven one
    minibatch at a time; and every epoch's index vector is a permutation of [0, B) that differs between epochs/updates."""
    g, meta = load("discrete_t128_n64_seed1")
    U = "u1/"
    B = meta["T"] * meta["N"]
    MB = B // meta["nmb"]
    ctx = make_ctx(P, meta)
    _load_batch(ctx, g, U, meta)
    ctx.set_params(g[U + "params_before"])
    ctx.set_learning_rate(1e-3)
    ctx.update()
    p_fused = ctx.get_params()
    perm = ctx.read("PERM", (meta["epochs"], B))
    for e in range(meta["epochs"]):
        assert np.array_equal(np.sort(perm[e]), np.arange(B))
    assert not np.array_equal(perm[0], perm[1]) and not np.array_equal(perm[0], np.arange(B))
    st_fused = ctx.stats()
    ctx2 = make_ctx(P, meta)
    _load_batch(ctx2, g, U, meta)
    ctx2.set_params(g[U + "params_before"])
    ctx2.set_learning_rate(1e-3)
    for e in range(meta["epochs"]):
        for s in range(meta["nmb"]):
            ctx2.minibatch_forward_backward(perm[e, s * MB:(s + 1) * MB])
            ctx2.optimizer_step()
    assert np.array_equal(bits(p_fused), bits(ctx2.get_params()))
    ev = O.explained_variance(g[U + "gae_returns"], g[U + "values"])
    assert abs(st_fused["explained_variance"] - ev) <= 1e-5
    assert st_fused["optimizer_steps"] == meta["epochs"] * meta["nmb"]
    # oracle on the same permutations
    net = O.Net.make(4, [2])
    hp = O.HParams(gamma=meta["gamma"], gae_lambda=meta["lam"], clip_coef=meta["clip"], ent_coef=meta["ent"], vf_coef=meta["vf"],
                   max_grad_norm=meta["mgn"], norm_adv=meta["norm_adv"], clip_vloss=meta["clip_vloss"])
    p = g[U + "params_before"].copy()
    m, v = np.zeros_like(p), np.zeros_like(p)
    k = 0
    obs = g[U + "obs"].reshape(B, 4)
    for e in range(meta["epochs"]):
        for s in range(meta["nmb"]):
            grads, _ = O.minibatch_grads(net, hp, p, obs, g[U + "actions"].reshape(B), g[U + "logprobs"].ravel(), g[U + "gae_advantages"].ravel(),
                                         g[U + "gae_returns"].ravel(), g[U + "values"].ravel(), perm[e, s * MB:(s + 1) * MB])
            grads, _ = O.clip_grad_norm(net, grads, hp.max_grad_norm)
            p, m, v = O.adamw_step(p, grads, m, v, 1e-3, k + 1)
            k += 1
    assert np.abs(p - p_fused).max() <= 2e-6
    ctx.close()
    ctx2.close()


def test_update_kernel_matches_oracle_at_headline_minibatch(P):
    """The matrix-core minibatch step (fp32 operands as pairs of fp16 terms on f16 MFMAs, gather from the packed records) against the
    CPU oracle (scalar fp32 fmaf chains, library tanhf) on a 131 072-row minibatch drawn from a real rollout of BASELINE configs[1]'s
    shape: loss scalars to 1e-5 relative (north_star), gradients to 5e-6 of the largest."""
    cfg = dict(num_envs=4096, num_steps=128, num_minibatches=4, update_epochs=1, seed=3, total_timesteps=4096 * 128 * 2)
    ctx = P.Context(P.make_config(**cfg))
    ctx.init_orthogonal(5)
    ctx.env_reset()
    ctx.rollout()
    ctx.calc_advantage()
    perm = ctx.generate_permutations()
    idx = perm[0, :131072]
    grads = ctx.minibatch_forward_backward(idx)
    st = ctx.stats()
    T, N = 128, 4096
    net = O.Net.make(4, [2])
    hp = O.HParams(gamma=0.98, gae_lambda=0.95, clip_coef=0.2, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, norm_adv=1, clip_vloss=1)
    g_ref, s_ref = O.minibatch_grads(net, hp, ctx.get_params(), ctx.read("OBS", (T * N, 4)), ctx.read("ACTIONS", (T * N,)).astype(np.float32),
                                     ctx.read("LOGPROBS"), ctx.read("ADVANTAGES"), ctx.read("RETURNS"), ctx.read("VALUES"), idx)
    for n, key in (("pg_loss", "pg_loss"), ("v_loss", "v_loss"), ("entropy_loss", "entropy_loss"), ("approx_kl", "approx_kl"),
                   ("clipfrac", "clipfrac_last"), ("loss", "loss")):
        assert abs(st[key] - s_ref[n]) <= 1e-5 * max(1.0, abs(s_ref[n])), (n, st[key], s_ref[n])
    assert np.abs(grads - g_ref).max() <= 5e-6 * np.abs(g_ref).max()
    ctx.close()


@pytest.mark.parametrize("scale", [1e-12, 1.0, 1e7])
def test_update_kernel_keeps_the_fp32_range(P, scale):
    """The matrix cores see fp32 operands as pairs of fp16 terms; the back-propagated gradient is kept inside fp16's range by a
    wave-uniform power-of-two factor that follows the data.  Advantages, returns and old values scaled by 1e-12 .. 1e7 (norm_adv off,
    so the factor reaches the gradient): the gradient still matches the fp32 oracle to 5e-6 of its largest element and the losses to
    1e-5 -- nothing overflows to inf and nothing underflows to zero."""
    cfg = dict(num_envs=256, num_steps=128, num_minibatches=1, update_epochs=1, seed=4, total_timesteps=256 * 128 * 2, norm_adv=0,
               ent_coef=0.01)
    ctx = P.Context(P.make_config(**cfg))
    ctx.init_orthogonal(6)
    ctx.env_reset()
    ctx.rollout()
    ctx.calc_advantage()
    T, N = 128, 256
    for name in ("ADVANTAGES", "RETURNS", "VALUES"):
        ctx.write(name, (ctx.read(name).astype(np.float64) * scale).astype(np.float32))
    idx = ctx.generate_permutations()[0]
    grads = ctx.minibatch_forward_backward(idx)
    st = ctx.stats()
    net = O.Net.make(4, [2])
    hp = O.HParams(gamma=0.98, gae_lambda=0.95, clip_coef=0.2, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, norm_adv=0, clip_vloss=1)
    g_ref, s_ref = O.minibatch_grads(net, hp, ctx.get_params(), ctx.read("OBS", (T * N, 4)), ctx.read("ACTIONS", (T * N,)).astype(np.float32),
                                     ctx.read("LOGPROBS"), ctx.read("ADVANTAGES"), ctx.read("RETURNS"), ctx.read("VALUES"), idx)
    assert np.isfinite(grads).all() and np.abs(g_ref).max() > 0
    for n, key in (("pg_loss", "pg_loss"), ("v_loss", "v_loss"), ("loss", "loss")):
        assert abs(st[key] - s_ref[n]) <= 1e-5 * max(abs(s_ref[n]), 1e-30) + 1e-12 * scale, (n, st[key], s_ref[n])
    shapes = ctx.param_shapes()
    off = 0
    for i, shp in enumerate(shapes):   # per tensor: a tensor of small gradients must not be lost beside a large one
        n = int(np.prod(shp))
        g, r = grads[off:off + n], g_ref[off:off + n]
        assert np.abs(g - r).max() <= 5e-6 * max(np.abs(r).max(), 1e-30), (i, shp, np.abs(g - r).max(), np.abs(r).max())
        off += n
    ctx.close()


def test_ragged_minibatches(P):
    """batch % num_minibatches != 0: minibatch_size is the integer quotient and a short extra minibatch follows (reference
    PPO_Discrete.cpp:247,573-576).  B = 256 in 3 minibatches -> 85, 85, 85, 1 rows.  The 85-row steps (two full 32-sample tiles and a
    partial one) must match the oracle like any other; the 1-row minibatch has no Bessel-corrected std (torch .std() of one element is
    NaN, :592-594), so its loss, its gradient and from then on every parameter are NaN -- in the reference, in the oracle and here."""
    g, meta = load("discrete_t32_n8_seed2")
    B, MB = 256, 85
    net = O.Net.make(meta["obs"], [meta["act"]])
    hp = O.HParams(gamma=meta["gamma"], gae_lambda=meta["lam"], clip_coef=meta["clip"], ent_coef=meta["ent"], vf_coef=meta["vf"],
                   max_grad_norm=meta["mgn"], norm_adv=meta["norm_adv"], clip_vloss=meta["clip_vloss"])
    U = "u1/"
    batch = (g[U + "obs"].reshape(B, -1), g[U + "actions"].reshape(B), g[U + "logprobs"].ravel(), g[U + "gae_advantages"].ravel(),
             g[U + "gae_returns"].ravel(), g[U + "values"].ravel())
    ctx = make_ctx(P, meta, num_minibatches=3, update_epochs=2)
    _load_batch(ctx, g, U, meta)
    ctx.set_params(g[U + "params_before"])
    ctx.set_learning_rate(1e-3)
    perm = np.random.default_rng(11).permutation(B).astype(np.int32)
    p = g[U + "params_before"].copy()
    m, v = np.zeros_like(p), np.zeros_like(p)
    for k in range(4):
        idx = perm[k * MB:min((k + 1) * MB, B)]
        assert len(idx) == (85 if k < 3 else 1)
        grads = ctx.minibatch_forward_backward(idx)
        st = ctx.stats()
        g_ref, ref = O.minibatch_grads(net, hp, p, *batch, idx)
        if k < 3:
            for n, key in (("pg_loss", "pg_loss"), ("v_loss", "v_loss"), ("approx_kl", "approx_kl"), ("clipfrac", "clipfrac_last"), ("loss", "loss")):
                assert abs(st[key] - ref[n]) <= 1e-5 * max(1.0, abs(ref[n])), (k, n, st[key], ref[n])
            assert np.abs(grads - g_ref).max() <= 1e-6 + 1e-4 * np.abs(g_ref).max(), k
        else:
            actor = slice(O.param_count(net) - (64 * meta["obs"] + 64 + 64 * 64 + 64 + meta["act"] * 64 + meta["act"]), None)
            assert np.isnan(g_ref[actor]).all() and np.isnan(grads[actor]).all()     # the policy gradient carries the NaN advantage
            assert np.isnan(st["pg_loss"]) and np.isnan(st["loss"]) and np.isfinite(st["v_loss"])
        ctx.optimizer_step()
        g_c, _ = O.clip_grad_norm(net, g_ref, hp.max_grad_norm)
        p, m, v = O.adamw_step(p, g_c, m, v, 1e-3, k + 1)
        got = ctx.get_params()
        if k < 3:
            assert np.abs(got - p).max() <= 2e-6, k
        else:
            assert np.isnan(got).all() and np.isnan(p).all()   # clip_grad_norm_ spreads the NaN norm to every parameter (:640)
    ctx.close()
    # and the fused loop walks the same four minibatches per epoch
    ctx = make_ctx(P, meta, num_minibatches=3, update_epochs=2)
    _load_batch(ctx, g, U, meta)
    ctx.set_params(g[U + "params_before"])
    ctx.update()
    assert ctx.stats()["optimizer_steps"] == 2 * 4
    assert np.isnan(ctx.get_params()).all()
    ctx.close()


# ------------------------------------------------------------------------------------------- end to end
@pytest.mark.parametrize("vector", [False, True])
@pytest.mark.parametrize("N", [96, 7, 33])   # 7, 33: the last 16-env tile of the fused rollout is ragged
def test_fused_rollout_equals_stepwise_api(P, N, vector):
    """ppo_rollout (one launch) against T x { policy_act, env_step } through the stand-alone entry points, FREE-RUNNING: the stepwise loop samples its own
    actions.  Everything must agree BIT FOR BIT -- observations, rewards, done flags, auto-resets, truncation, values, log-probs, and every sampled action.
    The stand-alone policy runs the arithmetic of the kernel the context's rollout runs (round 6; include/ppo_hip.h): by default rollout16_kernel's
    (matrix cores, 16 rows per tile, fp32 carried as two fp16 terms: policy_act16_kernel), under PPO_KERNEL_ROLLOUT_VECTOR the vector ALU's fp32
    multiply-adds (rollout2_kernel / policy_act_kernel) -- the rollout for golden replays against earlier rounds' recordings.  Until round 6 the stand-alone
    policy was always the vector form and a default context's free-running rollout could differ from it in <= 2 of 8 192 actions."""
    cfg = dict(num_envs=N, num_steps=40, num_minibatches=1, update_epochs=1, seed=5, max_episode_steps=30, kernel_flags=P.KERNEL_ROLLOUT_VECTOR if vector else 0)
    a = P.Context(P.make_config(**cfg))
    b = P.Context(P.make_config(**cfg))
    a.init_orthogonal(11)
    params = a.get_params()
    params[-130:] *= 30.0
    a.set_params(params)
    b.set_params(params)
    a.env_reset()
    obs = b.env_reset()
    a.rollout()
    T = 40
    r_obs, r_act, r_lp, r_v, r_rew, r_done = (a.read("OBS", (T, N, 4)), a.read("ACTIONS", (T, N)), a.read("LOGPROBS", (T, N)),
                                              a.read("VALUES", (T, N)), a.read("REWARDS", (T, N)), a.read("DONES", (T, N)))
    done = np.zeros(N, np.float32)
    for t in range(T):
        assert np.array_equal(bits(obs), bits(r_obs[t])) and np.array_equal(done, r_done[t])
        act, lp, en, v = b.policy_act(obs, step_index=t)                        # the stand-alone sampler on the same Philox word
        assert np.array_equal(act.ravel(), r_act[t]), (t, int((act.ravel() != r_act[t]).sum()))
        assert np.array_equal(bits(lp), bits(r_lp[t])) and np.array_equal(bits(v), bits(r_v[t]))
        f_act, f_lp, _, _ = b.policy_act(obs, action=act, step_index=t)        # teacher-forced with the same actions: the same log-prob bits
        assert np.array_equal(f_act, act) and np.array_equal(bits(f_lp), bits(lp))
        obs, rew, d = b.env_step(act)
        assert np.array_equal(rew, r_rew[t])
        done = d.astype(np.float32)
    assert np.array_equal(bits(obs), bits(a.read("NEXT_OBS", (N, 4))))
    assert r_done.sum() > 0  # truncation at 30 steps exercised
    a.close()
    b.close()


def test_the_two_rollout_arithmetics_agree_to_fp32_noise(P):
    """The matrix-core form (default) and the vector-ALU form (PPO_KERNEL_ROLLOUT_VECTOR) of the policy compute the same function: on the same observations and
    teacher-forced actions their log-probs agree to 1e-6 (3e-6 is the bar against the reference; 1 - 2 ULP measured), and of the actions each samples from the
    SAME Philox word at most a handful fall on different sides of a CDF edge.  The values come from one critic kernel in both: bit for bit."""
    N = 4096
    cfg = dict(num_envs=N, num_steps=4, num_minibatches=1, update_epochs=1, seed=5)
    m = P.Context(P.make_config(**cfg))
    v = P.Context(P.make_config(kernel_flags=P.KERNEL_ROLLOUT_VECTOR, **cfg))
    m.init_orthogonal(11)
    params = m.get_params()
    params[-130:] *= 30.0
    m.set_params(params)
    v.set_params(params)
    obs = np.random.default_rng(3).uniform(-0.3, 0.3, (2 * N + 5, 4)).astype(np.float32)
    am, lpm, enm, vm = m.policy_act(obs, step_index=9)
    av, lpv, env_, vv = v.policy_act(obs, step_index=9)
    assert int((am != av).sum()) <= 3
    _, lpm_f, enm_f, _ = m.policy_act(obs, action=av, step_index=9)
    np.testing.assert_allclose(lpm_f, lpv, rtol=0, atol=1e-6)
    np.testing.assert_allclose(enm_f, env_, rtol=0, atol=1e-6)
    assert np.array_equal(bits(vm), bits(vv))
    assert not np.array_equal(bits(lpm_f), bits(lpv))   # (they ARE two arithmetics: if this ever fails the flag has stopped selecting anything)
    m.close()
    v.close()


def test_policy_act_follows_the_rollout_into_the_vector_kernel_when_weights_leave_fp16(P):
    """An actor output-layer weight of 300 does not fit rollout16_kernel's fp16 operand: the rollout of a DEFAULT context takes the vector kernel for that
    launch (the test below), and so must the stand-alone policy -- same range snapshot, same decision -- or the two would disagree exactly where a drop-in
    user cannot see why.  Free-running, bit for bit, and counted in ppo_profile.vector_fallback_launches."""
    N, T = 48, 12
    cfg = dict(num_envs=N, num_steps=T, num_minibatches=1, update_epochs=1, seed=5)
    a = P.Context(P.make_config(**cfg))
    b = P.Context(P.make_config(**cfg))
    a.init_orthogonal(3)
    params = a.get_params()
    params[-130] = 300.0
    a.set_params(params)
    b.set_params(params)
    a.env_reset()
    obs = b.env_reset()
    a.rollout()
    r_act, r_lp = a.read("ACTIONS", (T, N)), a.read("LOGPROBS", (T, N))
    for t in range(T):
        act, lp, _, _ = b.policy_act(obs, step_index=t)
        assert np.array_equal(act.ravel(), r_act[t]) and np.array_equal(bits(lp), bits(r_lp[t]))
        obs, _, _ = b.env_step(act)
    assert a.profile_read()["vector_fallback_launches"] >= 1 and b.profile_read()["vector_fallback_launches"] >= T
    b.stats()   # no range error was raised
    a.close()
    b.close()


def test_weights_beyond_fp16_take_the_vector_kernels_for_that_launch(P):
    """The matrix-core kernels carry some operands as fp16 (rollout16_kernel: 2^8 W3, |W3| < 255; the update kernels: c W2 and the products through its
    columns).  The reference has no such limits, and a drop-in user must not meet them: a small kernel takes the maximum of |parameter| per class once per update (and after every host write),
    the host reads its pinned mirror, and a launch whose weights do not fit takes the vector kernel (plain fp32) -- with DEFAULT flags.  Here: an actor
    output-layer weight of 300, and a hidden-to-hidden weight of 6, each set through ppo_params_set_h: the rollout equals the PPO_KERNEL_ROLLOUT_VECTOR context's
    bit for bit (it IS that kernel), the update equals the PPO_KERNEL_UPDATE_VECTOR context's, training goes on and the statistics read reports nothing."""
    cfg = dict(num_envs=64, num_steps=16, num_minibatches=2, update_epochs=2, seed=5, total_timesteps=64 * 16 * 4)
    # actorOutputLayer.weight[0][0] against the context whose ROLLOUT is always the vector kernel; criticMiddleLayer.weight[0][7] against the one whose UPDATE is
    for where, value, twin_flags in ((-130, 300.0, P.KERNEL_ROLLOUT_VECTOR), (4 * 64 + 64 + 7, 6.0, P.KERNEL_UPDATE_VECTOR)):
        ctxs = [P.Context(P.make_config(kernel_flags=f, **cfg)) for f in (0, twin_flags)]
        ctxs[0].init_orthogonal(3)
        params = ctxs[0].get_params()
        params[where] = value
        outs = []
        for ctx in ctxs:
            ctx.set_params(params)
            ctx.env_reset()
            for _ in range(2):
                ctx.train_iteration()
            st = ctx.stats()                      # raises PPO_ERR_STATE if any kernel flagged a range error
            assert np.isfinite(st["loss"]) and st["optimizer_steps"] == 8
            outs.append((ctx.read("LOGPROBS"), ctx.read("ACTIONS"), ctx.read("VALUES"), ctx.get_params()))
            ctx.close()
        assert np.isfinite(outs[0][0]).all() and np.isfinite(outs[0][3]).all()
        # the default context took the vector kernel wherever its twin does: rollout buffers and parameters after two iterations, bit for bit
        assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][0].view(np.uint32), outs[1][0].view(np.uint32))
        assert np.array_equal(outs[0][3].view(np.uint32), outs[1][3].view(np.uint32))
        assert abs(outs[0][3][where] - value) < 0.1


def test_weights_that_grow_past_the_thresholds_by_optimizer_steps_alone(P):
    """The other way into the fp16-range fallback: nobody writes the parameters -- AdamW itself moves them past the thresholds (learning rate 1.0: a weight moves by up to
    the parameters pass the hidden-to-hidden threshold of 4 within a few updates).  The maxima are swept once per update on a stream of their own into a pinned mirror the host reads
    without synchronising, so the dispatch sees them an update late; the thresholds sit at half the kernels' limits for exactly that lag.  What must hold: no range error is ever raised, the
    run stays finite, and from some update on the launches are counted as vector fallbacks (`ppo_profile.vector_fallback_launches`)."""
    ctx = P.Context(P.make_config(num_envs=64, num_steps=16, num_minibatches=2, update_epochs=4, seed=11, total_timesteps=64 * 16 * 12, learning_rate=1.0, anneal_lr=False))
    ctx.init_orthogonal(2)
    ctx.env_reset()
    seen = []
    for it in range(10):
        ctx.train_iteration()
        st = ctx.stats()                       # raises PPO_ERR_STATE if any kernel flagged a range error
        p = ctx.get_params()
        assert np.isfinite(p).all() and np.isfinite(st["loss"]), it
        seen.append((float(np.abs(p).max()), ctx.profile_read()["vector_fallback_launches"]))
    assert seen[-1][0] > 4.0, seen            # the weights did grow past the hidden-to-hidden threshold
    assert seen[-1][1] > 0 and seen[0][1] <= seen[-1][1], seen   # ... and the launches took the vector kernels from some update on
    ctx.close()


def test_observation_beyond_fp16_is_reported(P):
    """The wave-specialised update kernel cuts the observation into fp16 terms: |obs| >= 65504 written through the C-ABI's OBS buffer does not fit.  The record
    packing checks it (PPO_ERRFLAG_UPDATE_RANGE) and the next statistics read fails with PPO_ERR_STATE naming PPO_KERNEL_UPDATE_VECTOR -- no silent NaN.
    The optimizer kernels read the same error word and do NOT apply a step the device knows is garbage (ppo_hip.h, ABI 5): parameters and AdamW moments after
    the update are the ones before it, bit for bit.  With PPO_KERNEL_UPDATE_VECTOR the same batch trains."""
    for flags, ok in ((0, False), (P.KERNEL_UPDATE_VECTOR, True)):
        ctx = P.Context(P.make_config(num_envs=32, num_steps=8, num_minibatches=2, update_epochs=2, seed=2, kernel_flags=flags))
        ctx.init_orthogonal(1)
        ctx.env_reset()
        ctx.train_iteration()                 # one clean iteration first: the moments are not all zero
        ctx.rollout()
        ctx.calc_advantage()
        before = [ctx.get_params(), ctx.read("EXP_AVG"), ctx.read("EXP_AVG_SQ")]
        obs = ctx.read("OBS", (8, 32, 4))
        obs[3, 5, 1] = 1.0e5
        ctx.write("OBS", obs)
        ctx.update()
        after = [ctx.get_params(), ctx.read("EXP_AVG"), ctx.read("EXP_AVG_SQ")]
        if ok:
            assert np.isfinite(ctx.stats()["loss"]) and np.isfinite(after[0]).all()
            assert not np.array_equal(before[0], after[0])
        else:
            for b, a in zip(before, after):
                assert np.array_equal(b.view(np.uint32), a.view(np.uint32))      # four optimizer steps ran, none was applied
            with pytest.raises(P.binding.PPOError, match="PPO_KERNEL_UPDATE_VECTOR"):
                ctx.stats()
        ctx.close()


def test_observation_beyond_fp16_is_no_error_where_the_vector_kernel_runs(P):
    """The observation's fp16 range is a limit of the wave-specialised matrix-core kernel only.  In an update whose WEIGHTS already send every launch to the vector
    kernel (default flags, a hidden-to-hidden weight of 6: test_weights_beyond_fp16_take_the_vector_kernels_for_that_launch) an observation of 1e5 is not an error:
    the update trains, the statistics read reports nothing, and the launches are counted in ppo_profile.vector_fallback_launches."""
    ctx = P.Context(P.make_config(num_envs=32, num_steps=8, num_minibatches=2, update_epochs=2, seed=2))
    ctx.init_orthogonal(1)
    params = ctx.get_params()
    params[4 * 64 + 64 + 7] = 6.0
    ctx.set_params(params)
    ctx.env_reset()
    ctx.rollout()
    ctx.calc_advantage()
    obs = ctx.read("OBS", (8, 32, 4))
    obs[3, 5, 1] = 1.0e5
    ctx.write("OBS", obs)
    ctx.update()
    st = ctx.stats()
    assert np.isfinite(st["loss"]) and np.isfinite(ctx.get_params()).all() and not np.array_equal(ctx.get_params(), params)
    assert ctx.profile_read()["vector_fallback_launches"] == 4
    ctx.close()


@pytest.mark.parametrize("T,limit", [(1, 500), (2, 1), (5, 1), (7, 2), (64, 3), (33, 500)])
def test_fused_rollout_at_constant_resets_and_short_horizons(P, T, limit):
    """The fused rollout forms the NEXT step's sin / cos a step ahead of the action (CartPole's next pose does not depend on it) with its own copy of
    the episode length and the reset count, hands reset rows over through LDS and defers a step's stores into the next step: drive exactly those
    corners -- every step (or every second / third) ends an episode by truncation, one- and two-step rollouts, two rollouts back to back (the
    counters persist) -- and compare the env side bit for bit with the stand-alone step kernel driven by the rollout's own actions."""
    N = 40
    cfg = dict(num_envs=N, num_steps=T, num_minibatches=1, update_epochs=1, seed=9, max_episode_steps=limit)
    a, b = P.Context(P.make_config(**cfg)), P.Context(P.make_config(**cfg))
    a.init_orthogonal(3)
    b.set_params(a.get_params())
    a.env_reset()
    obs = b.env_reset()
    done = np.zeros(N, np.float32)
    for rollout in range(2):
        a.rollout()
        r_obs, r_act, r_rew, r_done = a.read("OBS", (T, N, 4)), a.read("ACTIONS", (T, N)), a.read("REWARDS", (T, N)), a.read("DONES", (T, N))
        fin_len = a.read("FIN_LEN", (T, N))
        for t in range(T):
            assert np.array_equal(bits(obs), bits(r_obs[t])) and np.array_equal(done, r_done[t]), (rollout, t)
            obs, rew, d = b.env_step(r_act[t].reshape(N, 1).astype(np.int64))
            assert np.array_equal(rew, r_rew[t])
            done = d.astype(np.float32)
            if limit <= 3:
                assert np.all(fin_len[t][d != 0] <= limit) and np.all(fin_len[t][d == 0] == 0)
        assert np.array_equal(bits(obs), bits(a.read("NEXT_OBS", (N, 4)))) and np.array_equal(done, a.read("NEXT_DONE").astype(np.float32))
    if limit == 1:
        assert r_done[1:].min() == 1.0 if T > 1 else True     # every step ends an episode
    a.close()
    b.close()


@pytest.mark.parametrize("kind", ["cartpole", "mountaincar", "nstep", "odd_envs"])
def test_train_iteration_equals_rollout_scan_update(P, kind):
    """ppo_train_iteration against its three parts called one after the other (rollout, calc_advantage, update): same seeds, same arithmetic --
    advantages, returns and parameters bit-identical, explained variance and loss equal.  (The iteration skips the stand-alone scan's
    critic call: the rollout's epilogue already left NEXT_VALUE.)"""
    kw = dict(num_envs=64, num_steps=32, num_minibatches=2, update_epochs=2, seed=8, total_timesteps=64 * 32 * 4, anneal_lr=False)
    if kind == "mountaincar":
        kw.update(env_kind=P.ENV_MOUNTAINCAR, dist_kind=P.DIST_MASKED, obs_size=2, head_dims=(3,), max_episode_steps=200)
    if kind == "nstep":
        kw.update(use_gae=False)
    if kind == "odd_envs":
        kw.update(num_envs=20, total_timesteps=20 * 32 * 4)
    a, b = P.Context(P.make_config(**kw)), P.Context(P.make_config(**kw))
    for c in (a, b):
        c.init_orthogonal(4)
        c.env_reset()
    for _ in range(2):
        a.train_iteration()
        b.rollout()
        b.calc_advantage()
        b.update()
    for name in ("ADVANTAGES", "RETURNS", "VALUES", "LOGPROBS"):
        assert np.array_equal(bits(a.read(name)), bits(b.read(name))), name
    assert np.array_equal(bits(a.get_params()), bits(b.get_params()))
    sa, sb = a.stats(), b.stats()
    assert abs(sa["explained_variance"] - sb["explained_variance"]) <= 1e-9 * max(1.0, abs(sb["explained_variance"]))
    assert sa["loss"] == sb["loss"]
    a.close(); b.close()


@pytest.mark.parametrize("kind", ["cartpole", "mountaincar"])
def test_headline_size_runs_are_bit_reproducible(P, kind):
    """BASELINE configs[1] / configs[3] sizes, twice from the same seed: every buffer and every parameter bit-identical after two iterations.
    The 256-workgroup update kernel adds its waves and workgroups in a fixed order (no float atomics), the permutations and the sampling are
    counter-based: nothing in the path depends on scheduling."""
    kw = dict(num_envs=4096, num_steps=128, num_minibatches=4, update_epochs=10, seed=21, total_timesteps=4096 * 128 * 4)
    if kind == "mountaincar":
        kw.update(env_kind=P.ENV_MOUNTAINCAR, dist_kind=P.DIST_MASKED, obs_size=2, head_dims=(3,), max_episode_steps=200, num_envs=8192,
                  total_timesteps=8192 * 128 * 4)
    runs = []
    for _ in range(2):
        c = P.Context(P.make_config(**kw))
        c.init_orthogonal(3)
        c.env_reset()
        for _ in range(2):
            c.train_iteration()
        runs.append((c.get_params(), c.read("ADVANTAGES"), c.read("LOGPROBS"), c.read("ACTIONS"), c.stats()))
        c.close()
    for a, b in zip(runs[0][:4], runs[1][:4]):
        assert np.array_equal(bits(a) if a.dtype == np.float32 else a, bits(b) if b.dtype == np.float32 else b)
    assert runs[0][4]["loss"] == runs[1][4]["loss"] and runs[0][4]["explained_variance"] == runs[1][4]["explained_variance"]


def test_training_learns_cartpole(P):
    """De-facto acceptance test of the reference (README.md:169-178): ep_len_mean climbs.  256 envs x 128 steps x 25 updates."""
    ctx = P.Context(P.make_config(num_envs=256, num_steps=128, num_minibatches=4, update_epochs=4, seed=2, total_timesteps=256 * 128 * 25,
                                  ent_coef=0.0, learning_rate=1e-3))
    ctx.init_orthogonal(2)
    ctx.env_reset()
    first = None
    for u in range(25):
        ctx.train_iteration()
        st = ctx.stats()
        assert np.isfinite(st["loss"])
        if first is None:
            first = st["ep_len_mean"]
    assert st["updates"] == 25 and st["global_step"] == 256 * 128 * 25
    assert st["ep_len_mean"] > max(100.0, 3 * first), (first, st)
    assert abs(st["learning_rate"] - 1e-3 * (1 - 24 / 25)) < 1e-9    # linear anneal, PPO_Discrete.cpp:515-517
    ctx.close()


def test_mountaincar_masked_iteration(P):
    ctx = P.Context(P.make_config(env_kind=P.ENV_MOUNTAINCAR, dist_kind=P.DIST_MASKED, obs_size=2, head_dims=(3,), num_envs=128, num_steps=64,
                                  max_episode_steps=200, seed=1, total_timesteps=128 * 64 * 4, ent_coef=0.01, gamma=0.99))
    ctx.init_orthogonal(3)
    obs = ctx.env_reset()
    assert np.all((obs[:, 0] >= -0.6) & (obs[:, 0] <= -0.4)) and np.all(obs[:, 1] == 0)
    assert len(np.unique(obs[:, 0])) > 100     # per-env keyed reset noise
    for _ in range(4):
        ctx.train_iteration()
    st = ctx.stats()
    # true entropy on the masked path (CategoricalMasked.cpp:127-144): starts at ln 3 and stays a real entropy, never the
    # -FLT_MIN of the plain Categorical's clamp bug
    assert np.isfinite(st["loss"]) and 0.3 < st["entropy_loss"] <= np.log(3) + 1e-4
    assert np.all(ctx.read("MASKS") == 1)
    assert np.all(ctx.read("REWARDS") == -1.0)
    ctx.close()


def test_errors_are_reported_like_the_reference(P):
    with pytest.raises(P.binding.PPOError, match="The environment returned an observation of size 4, but your config defined"):
        P.Context(P.make_config(obs_size=2))
    with pytest.raises(P.binding.PPOError, match="only the reference architecture"):
        P.Context(P.make_config(hidden=256, n_hidden=4))


# ------------------------------------------------------------------------------------------- data parallel
def test_sharded_rollout_equals_columns_of_the_global_rollout(P):
    """Env sharding: RNG streams, the shared reset stream and the 'env 0 is reset twice' quirk are tied to GLOBAL env indices,
    so shard r of a 2-way split reproduces columns [r*N/2, (r+1)*N/2) of the single-context rollout bit for bit."""
    T, N = 64, 128
    base = dict(num_steps=T, num_minibatches=4, update_epochs=1, seed=9, max_episode_steps=40, total_timesteps=T * N * 2)
    whole = P.Context(P.make_config(num_envs=N, **base))
    whole.init_orthogonal(4)
    params = whole.get_params()
    params[-130:] *= 25.0
    whole.set_params(params)
    whole.env_reset()
    whole.rollout()
    for r in range(2):
        sh = P.Context(P.dist.shard_config(P.make_config, r, 2, N, **base))
        sh.set_params(params)
        sh.env_reset()
        sh.rollout()
        sl = slice(r * N // 2, (r + 1) * N // 2)
        for name, shape in (("OBS", (T, N, 4)), ("ACTIONS", (T, N)), ("LOGPROBS", (T, N)), ("REWARDS", (T, N)), ("DONES", (T, N)), ("VALUES", (T, N))):
            full = whole.read(name, shape)
            part = sh.read(name, (T, N // 2) + shape[2:])
            assert np.array_equal(full[:, sl].view(np.uint32) if full.dtype == np.float32 else full[:, sl], part.view(np.uint32) if part.dtype == np.float32 else part), (r, name)
        assert np.array_equal(bits(whole.read("NEXT_OBS", (N, 4))[sl]), bits(sh.read("NEXT_OBS", (N // 2, 4))))
        sh.close()
    whole.close()


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_update_equals_single_context(P, world):
    """2 and 8 contexts (one host thread each, all on this GPU) joined by the in-process communicator run the same protocol as the RCCL path:
    advantage sums of the global minibatch, then ONE all-reduce of the 1/M_global-scaled gradient per optimizer step.
    Losses, gradient and parameters equal the single-context step on the concatenated minibatch."""
    import threading
    g, meta = load("discrete_t128_n64_seed1")
    T, N, U = meta["T"], meta["N"], "u1/"
    rng = np.random.default_rng(3)
    steps = []
    for _ in range(3):
        t_rows = rng.choice(T, 32, replace=False)
        steps.append(np.array([t * N + e for t in t_rows for e in range(N)], np.int32))
    # single context
    one = make_ctx(P, meta)
    _load_batch(one, g, U, meta)
    one.set_params(g[U + "params_before"])
    one.set_learning_rate(1e-3)
    ref = []
    for rows in steps:
        grads = one.minibatch_forward_backward(rows)
        st = one.stats()
        one.optimizer_step()
        ref.append((grads, st, one.get_params()))
    one.close()
    # `world` ranks
    out = [None] * world
    errors = []

    def run(rank):
        try:
            n, off = P.dist.shard_envs(N, rank, world)
            ctx = make_ctx(P, meta, num_envs=n, env_offset=off, global_num_envs=N)
            ctx.comm_init_local(1234 + world, rank, world)
            sl = slice(off, off + n)
            ctx.write("OBS", np.ascontiguousarray(g[U + "obs"][:, sl]))
            ctx.write("ACTIONS", np.ascontiguousarray(g[U + "actions"].reshape(T, N, 1)[:, sl].astype(np.int32)))
            for name, key in (("LOGPROBS", "logprobs"), ("VALUES", "values"), ("ADVANTAGES", "gae_advantages"), ("RETURNS", "gae_returns")):
                ctx.write(name, np.ascontiguousarray(g[U + key][:, sl]))
            ctx.set_params(g[U + "params_before"])
            ctx.set_learning_rate(1e-3)
            res = []
            for rows in steps:
                local = np.array(P.dist.local_rows_of_global_rows(rows, T, N, rank, world), np.int32)
                d = ctx.dev(local, np.int32)
                P.binding._check(P.binding.lib().ppo_minibatch_forward_backward(ctx.h, d.ptr, __import__("ctypes").c_int64(local.size)), ctx.h)
                P.binding._check(P.binding.lib().ppo_allreduce_grads(ctx.h), ctx.h)
                grads = ctx.read("GRADS")
                ctx.optimizer_step()
                res.append((grads, ctx.stats(), ctx.get_params()))
            out[rank] = res
            ctx.close()
        except Exception as ex:  # surface failures of the worker threads
            errors.append(ex)

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errors, errors
    assert all(o is not None for o in out)
    for k, (grads, st, params) in enumerate(ref):
        for r in range(world):
            g2, st2, p2 = out[r][k]
            assert np.abs(g2 - grads).max() <= 2e-6 * max(1.0, np.abs(grads).max()), (k, r)
            for key in ("pg_loss", "v_loss", "entropy_loss", "approx_kl", "clipfrac_last", "loss", "total_norm"):
                assert abs(st2[key] - st[key]) <= 2e-6 * max(1.0, abs(st[key])), (k, r, key, st2[key], st[key])
            assert np.abs(p2 - params).max() <= 2e-6, (k, r)
        assert np.array_equal(bits(out[0][k][2]), bits(out[1][k][2]))   # replicas stay bit-identical


@pytest.mark.parametrize("world,N,T,max_steps", [(2, 128, 64, 40), (8, 128, 64, 40), (4, 16, 32, 25)])
def test_sharded_job_statistics_equal_single_context(P, world, N, T, max_steps):
    """ppo_read_stats of a sharded run is the JOB's table, identical on every rank (the reference prints one: PPO_Discrete.cpp:700-774).  Every
    rank's ring of finished episodes, tagged with the episodes' positions in the reference's push order (step, then global env index; :474-480),
    and its explained-variance sums (:647-648) ride the per-update all-reduce; the host rebuilds the job's CircularBuffer(100).  After the FIRST
    iteration the job's rollout is, column for column, the single context's (same weights, global RNG streams), so the statistics must be the
    single context's: episode means exactly (more than 100 finished episodes in the first two shapes, fewer in the third), explained variance to
    float noise.  After the second iteration (permutations are per shard) they must still be the same on every rank."""
    import threading
    base = dict(num_steps=T, num_minibatches=2, update_epochs=1, seed=9, max_episode_steps=max_steps, total_timesteps=T * N * 4)
    whole = P.Context(P.make_config(num_envs=N, **base))
    whole.init_orthogonal(4)
    params = whole.get_params()
    whole.env_reset()
    whole.train_iteration()
    w1 = whole.stats()
    whole.close()
    keys = ("ep_len_mean", "ep_rew_mean", "ep_count", "explained_variance", "global_step", "loss", "approx_kl", "clipfrac_mean")
    out, errors = [None] * world, []

    def run(rank):
        try:
            ctx = P.Context(P.dist.shard_config(P.make_config, rank, world, N, **base))
            ctx.comm_init_local(4321 + world, rank, world)
            ctx.set_params(params)
            ctx.env_reset()
            ctx.train_iteration()
            s1 = ctx.stats()
            ctx.train_iteration()
            s2 = ctx.stats()
            out[rank] = ({k: s1[k] for k in keys}, {k: s2[k] for k in keys})
            ctx.close()
        except Exception as ex:
            errors.append(ex)

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errors, errors
    assert all(o is not None for o in out)
    for r in range(1, world):
        assert out[r] == out[0], r
    s1 = out[0][0]
    assert s1["ep_count"] == w1["ep_count"] and (s1["ep_count"] == 100) == (N > 16)
    assert s1["ep_len_mean"] == w1["ep_len_mean"] and s1["ep_rew_mean"] == w1["ep_rew_mean"]
    assert s1["global_step"] == w1["global_step"] == N * T
    assert abs(s1["explained_variance"] - w1["explained_variance"]) <= 2e-6


def _rccl_selftest_ctx(P, monkeypatch, cfg, selftest):
    ctx = P.Context(P.make_config(kernel_flags=P.KERNEL_COMM_SELFTEST if selftest else 0, **cfg))
    if selftest:
        ctx.comm_init(P.comm_unique_id(), 0, 1)
    return ctx


@pytest.mark.parametrize("epochs,nmb,iters", [(1, 1, 1), (2, 4, 2)])
def test_rccl_single_rank_selftest(P, monkeypatch, epochs, nmb, iters):
    """The RCCL calls of the multi-rank path on ONE GPU: with PPO_KERNEL_COMM_SELFTEST in ppo_config.kernel_flags a one-rank communicator is really created
    (ncclGetUniqueId / ncclCommInitRank from the dlopen'ed librccl) and every collective of an update really goes through
    ncclAllReduce (f32 gradient + loss-sum tail per optimizer step, f64 statistics block + advantage sums per update), followed by the
    three-kernel optimizer path the ranks of a multi-GPU job take.  Sums over one rank are the identity, so the run must reproduce the
    plain single-context run to float noise.  Both contexts are driven with the SAME forced actions (the env is deterministic given
    them), so a sampled action cannot flip on a 1e-7 difference of a logit and the 16-step case stays a numerics check: parameters within
    2e-5 (measured ~2e-6: the two optimizer paths add the squares of the gradient in different orders and AdamW divides by sqrt(v))."""
    cfg = dict(num_envs=256, num_steps=64, num_minibatches=nmb, update_epochs=epochs, seed=7, total_timesteps=256 * 64 * 4)
    rng = np.random.default_rng(11)
    forced = [rng.integers(0, 2, size=(64, 256, 1)).astype(np.int64) for _ in range(iters)]

    def run(selftest):
        ctx = _rccl_selftest_ctx(P, monkeypatch, cfg, selftest)
        ctx.init_orthogonal(7)
        ctx.env_reset()
        for k in range(iters):
            ctx.rollout(forced[k])
            ctx.calc_advantage()
            ctx.update()
        out = (ctx.get_params(), ctx.stats())
        ctx.close()
        return out

    p0, s0 = run(False)
    p1, s1 = run(True)
    assert np.all(np.isfinite(p1))
    assert s1["optimizer_steps"] == s0["optimizer_steps"] == iters * epochs * nmb
    assert np.abs(p1 - p0).max() <= 2e-5, np.abs(p1 - p0).max()
    for key in ("pg_loss", "v_loss", "loss", "approx_kl", "total_norm", "explained_variance", "ep_len_mean", "ep_rew_mean"):
        assert abs(s1[key] - s0[key]) <= 2e-5 * max(1.0, abs(s0[key])), (key, s1[key], s0[key])
    assert s1["ep_count"] == s0["ep_count"]


def test_rccl_single_rank_selftest_free_running(P, monkeypatch):
    """The same path with its own sampled rollouts, two iterations: finiteness and bookkeeping only (sampled actions may flip on float noise)."""
    cfg = dict(num_envs=256, num_steps=64, num_minibatches=4, update_epochs=2, seed=7, total_timesteps=256 * 64 * 4)
    ctx = _rccl_selftest_ctx(P, monkeypatch, cfg, True)
    ctx.init_orthogonal(7)
    ctx.env_reset()
    for _ in range(2):
        ctx.train_iteration()
    st = ctx.stats()
    assert np.all(np.isfinite(ctx.get_params())) and np.isfinite(st["loss"]) and st["optimizer_steps"] == 16
    ctx.close()


def test_orthogonal_init_statistics(P):
    """ppo_params_init_orthogonal = Agent::ppoLayerInit (Agent.cpp:91-99; gains :25-37): torch::nn::init::orthogonal_(W, gain) leaves
    W W^T = gain^2 I when W has no more rows than columns and W^T W = gain^2 I otherwise, constant_(bias, 0).  Gains: sqrt(2) on hidden
    layers, 1.0 on the critic head, 0.01 on the actor head.  (The entries come from the build's own Gaussian, not LibTorch's: the
    property, not the bits, is what is checked.)  Also: different seeds give different matrices, the same seed the same."""
    ctx = P.Context(P.make_config(num_envs=8, num_steps=4, num_minibatches=1, update_epochs=1))
    ctx.init_orthogonal(11)
    p = ctx.get_params()
    shapes = ctx.param_shapes()
    assert [tuple(s) for s in shapes] == [(64, 4), (64, 1), (64, 64), (64, 1), (1, 64), (1, 1), (64, 4), (64, 1), (64, 64), (64, 1), (2, 64), (2, 1)]
    gains = [2 ** 0.5, None, 2 ** 0.5, None, 1.0, None, 2 ** 0.5, None, 2 ** 0.5, None, 0.01, None]
    off = 0
    for (r, c), gain in zip(shapes, gains):
        w = p[off:off + r * c].astype(np.float64)
        off += r * c
        if gain is None:
            assert np.all(w == 0.0)     # bias
            continue
        w = w.reshape(r, c)
        gram = w @ w.T if r <= c else w.T @ w
        np.testing.assert_allclose(gram, gain * gain * np.eye(min(r, c)), rtol=0, atol=2e-6 * gain * gain + 1e-12)
    assert off == p.size
    ctx.init_orthogonal(11)
    assert np.array_equal(bits(ctx.get_params()), bits(p))
    ctx.init_orthogonal(12)
    assert not np.array_equal(bits(ctx.get_params()), bits(p))
    ctx.close()
    # a generic network (configs[4] shape): every hidden layer sqrt(2), heads 1.0 / 0.01
    ctx = P.Context(P.make_config(env_kind=P.ENV_SYNTHETIC, dist_kind=P.DIST_MASKED, obs_size=376, head_dims=(3, 3, 3, 2), hidden=256, n_hidden=4,
                                  num_envs=8, num_steps=4, num_minibatches=1, update_epochs=1))
    ctx.init_orthogonal(3)
    p = ctx.get_params()
    shapes = [tuple(s) for s in ctx.param_shapes()]
    off = 0
    for i, (r, c) in enumerate(shapes):
        w = p[off:off + r * c].astype(np.float64)
        off += r * c
        if c == 1 and i % 2 == 1:
            assert np.all(w == 0.0)
            continue
        w = w.reshape(r, c)
        head = r in (1, 11)
        gain = (1.0 if r == 1 else 0.01) if head else 2 ** 0.5
        gram = w @ w.T if r <= c else w.T @ w
        np.testing.assert_allclose(gram, gain * gain * np.eye(min(r, c)), rtol=0, atol=2e-6 * gain * gain + 1e-12)
    assert off == p.size
    ctx.close()


def test_stats_snapshots_are_read_in_order_without_draining(P):
    """ppo_stats_snapshot / ppo_stats_snapshot_read (ppo_hip.h): a host that prints a table per update (printPPOResults, PPO_Discrete.cpp:700-774) runs one
    iteration ahead -- snapshot k, enqueue iteration k + 1 and its snapshot, THEN read snapshot k.  Snapshot k must hold iteration k's statistics (not
    k + 1's: learning rate, step counters, losses, episode means), two may be pending, a third is refused, and ppo_read_stats refuses to jump the queue."""
    cfg = dict(num_envs=256, num_steps=64, num_minibatches=4, update_epochs=2, seed=7, max_episode_steps=30, total_timesteps=256 * 64 * 4)
    ref = P.Context(P.make_config(**cfg))
    ref.init_orthogonal(7)
    ref.env_reset()
    want = []
    for _ in range(3):
        ref.train_iteration()
        want.append(ref.stats())
    ref.close()
    ctx = P.Context(P.make_config(**cfg))
    ctx.init_orthogonal(7)
    ctx.env_reset()
    ctx.train_iteration()
    ctx.stats_snapshot()
    ctx.train_iteration()
    ctx.stats_snapshot()
    with pytest.raises(P.binding.PPOError, match="two statistics snapshots are pending"):
        ctx.stats_snapshot()
    with pytest.raises(P.binding.PPOError, match="snapshot is pending"):
        ctx.stats()
    got1 = ctx.stats_snapshot_read()
    ctx.train_iteration()          # iteration 3 is enqueued before snapshot 2 is read
    ctx.stats_snapshot()
    got2 = ctx.stats_snapshot_read()
    got3 = ctx.stats_snapshot_read()
    assert [got1, got2, got3] == want
    assert got1["updates"] == 1 and got2["updates"] == 2 and got1["learning_rate"] > got2["learning_rate"] > got3["learning_rate"]
    with pytest.raises(P.binding.PPOError, match="no statistics snapshot is pending"):
        ctx.stats_snapshot_read()
    assert ctx.stats() == want[2]
    ctx.close()
