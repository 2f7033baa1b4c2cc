"""Pins the CPU oracle (oracle/ppo_oracle.c) against golden vectors produced by the compiled reference itself
(oracle/ref_harness.cpp -> tests/golden/*.pgld; every train scenario there is certified bit-identical to the
reference's own PPO_Discrete::train()).  CPU-only.

Tolerances: bit-exact for env transitions, reset stream, GAE and the AdamW element-wise step (given identical
gradients); 1e-5 (north_star's fp32 tolerance) or tighter for everything that goes through LibTorch reductions,
tanh/exp/log (Sleef vs libm differ by ULPs) -- the tolerance used is written next to each assertion.
"""
import os

import numpy as np
import pytest

import oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DISCRETE = ["discrete_t32_n8_seed2", "discrete_t64_n16_seed3_trunc", "discrete_t128_n64_seed1",
            # the shapes the reference ships and BASELINE.json configs[0] names, as they are (oracle/ref_harness.cpp: golden_shipped)
            "discrete_shipped_toml_t32_n8_act1", "discrete_config0_t128_n8_seed2"]
MASKED = ["multidiscrete_mountaincar_t32_n16"]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def load(name):
    g = O.read_pgld(os.path.join(G, name + ".pgld"))
    m = g["meta"]
    meta = dict(T=int(m[0]), N=int(m[1]), obs=int(m[2]), act=int(m[3]), nmb=int(m[4]), epochs=int(m[5]),
                max_steps=int(m[6]), seed=int(m[7]), updates=int(m[8]), anneal=int(m[9]), use_gae=int(m[10]),
                norm_adv=int(m[11]), clip_vloss=int(m[12]), masked=int(m[13]))
    h = g["hparams"]
    hp = O.HParams(gamma=h[1], gae_lambda=h[2], clip_coef=h[3], ent_coef=h[4], vf_coef=h[5], max_grad_norm=h[6],
                   norm_adv=meta["norm_adv"], clip_vloss=meta["clip_vloss"])
    meta["lr"] = float(h[0])
    net = O.Net.make(meta["obs"], [meta["act"]], dist_kind=O.DIST_MASKED if meta["masked"] else O.DIST_CATEGORICAL)
    return g, meta, hp, net


def test_certified_against_reference_train():
    for name in DISCRETE:
        g, *_ = load(name)
        assert int(g["certified_bitwise"][0]) == 1, name
        assert np.array_equal(bits(g["params_after_reference_train"]), bits(g["u%d/params_after" % load(name)[1]["updates"]]))


def test_reset_stream_bit_exact():
    rs = O.read_pgld(os.path.join(G, "cartpole_reset_stream.pgld"))
    for k, v in rs.items():
        assert np.array_equal(bits(O.cartpole_reset_stream(int(k[4:]), v.shape[0])), bits(v)), k


def test_libm_restatement_matches_host_libm():
    # dense sweep of the range CartPole/MountainCar reach (|theta| < pi/4 branch and the reduce_fast branch)
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-0.8, 0.8, 20000), rng.uniform(-4, 4, 20000), [0.0, 1e-5, -1e-5, 0.20943952]]).astype(np.float32)
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.sinf.restype = ctypes.c_float
    libm.sinf.argtypes = [ctypes.c_float]
    libm.cosf.restype = ctypes.c_float
    libm.cosf.argtypes = [ctypes.c_float]
    hs = np.array([libm.sinf(float(v)) for v in x], np.float32)
    hc = np.array([libm.cosf(float(v)) for v in x], np.float32)
    assert np.array_equal(bits(O.sinf(x)), bits(hs))
    assert np.array_equal(bits(O.cosf(x)), bits(hc))


def test_cartpole_transitions_bit_exact():
    c = O.read_pgld(os.path.join(G, "cartpole_transitions.pgld"))
    ns, r, t = O.cartpole_step(c["state"], c["action"])
    assert np.array_equal(bits(ns), bits(c["next_state"]))
    assert np.array_equal(r, c["reward"]) and np.array_equal(t, c["terminated"])
    assert 1000 < int(c["terminated"].sum()) < 7000  # the fixture exercises both outcomes


def test_mountaincar_transitions_bit_exact():
    c = O.read_pgld(os.path.join(G, "mountaincar_transitions.pgld"))
    ns, r, t = O.mountaincar_step(c["state"], c["action"])
    assert np.array_equal(bits(ns), bits(c["next_state"]))
    assert np.array_equal(r, c["reward"]) and np.array_equal(t, c["terminated"])


def test_distributions():
    d = O.read_pgld(os.path.join(G, "distributions.pgld"))
    for name in ["cat1", "cat2", "cat3", "cat6", "masked2", "masked3", "masked6"]:
        kind = O.DIST_MASKED if name.startswith("masked") else O.DIST_CATEGORICAL
        res = O.categorical(kind, d[name + "/logits"], d.get(name + "/mask"), d[name + "/value"])
        for f in ["m_logits", "m_probs", "log_prob", "entropy"]:
            ref = d[name + "/" + f]
            fin = np.isfinite(ref) & (np.abs(ref) < 1e7)  # masked rows hold -1e8-ish log-probs
            np.testing.assert_allclose(res[f][fin], ref[fin], rtol=2e-6, atol=1e-6, err_msg=name + "/" + f)
        if kind == O.DIST_CATEGORICAL:
            # the reference's clamp bug: entropy == -FLT_MIN * sum(p) (Categorical.cpp:112-119)
            assert np.all(np.abs(d[name + "/entropy"]) < 2e-38) and np.all(np.abs(res["entropy"]) < 2e-38)


def test_multihead_masked_agent():
    g = O.read_pgld(os.path.join(G, "multihead_agent.pgld"))
    heads = [int(h) for h in g["heads"]]
    net = O.Net.make(g["x"].shape[1], heads, dist_kind=O.DIST_MASKED)
    assert O.param_count(net) == g["params"].size
    lp, en, v = O.evaluate(net, g["params"], g["x"], g["action_hn"].T, g["mask"])
    np.testing.assert_allclose(lp, g["logprob"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(en, g["entropy"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(v, g["value"].ravel(), rtol=1e-5, atol=2e-6)
    assert np.array_equal(g["action_out"], g["action_hn"].T)


@pytest.mark.parametrize("name", DISCRETE + MASKED)
def test_gae_bit_exact(name):
    g, meta, hp, net = load(name)
    for u in range(1, meta["updates"] + 1):
        U = "u%d/" % u
        adv, ret = O.gae(g[U + "rewards"], g[U + "values"], g[U + "dones"], g[U + "next_value"], g[U + "next_done"],
                         hp.gamma, hp.gae_lambda)
        assert np.array_equal(bits(adv), bits(g[U + "gae_advantages"])), name
        assert np.array_equal(bits(ret), bits(g[U + "gae_returns"])), name


@pytest.mark.parametrize("name", DISCRETE + MASKED)
def test_reference_nstep_branch_throws(name):
    # PPO_Discrete.cpp:318,324 assigns a [1,N] tensor into an [N] row: the reference's use_gae=false path cannot run.
    g, *_ = load(name)
    assert int(g["u1/nstep_branch_throws"][0]) == 1


@pytest.mark.parametrize("name", DISCRETE)
def test_vecenv_trace_bit_exact(name):
    """initEnvs + stepEnvs (auto-reset, truncation, shared reset stream, env 0 double reset) replayed with the
    reference's sampled actions reproduce the reference's obs/reward/done buffers bit for bit."""
    g, meta, hp, net = load(name)
    env = O.VecEnv(O.ENV_CARTPOLE, meta["N"], meta["seed"], meta["max_steps"])
    obs = env.init()
    assert np.array_equal(bits(obs), bits(g["init_obs"]))
    done = np.zeros(meta["N"], np.int32)
    for u in range(1, meta["updates"] + 1):
        U = "u%d/" % u
        for t in range(meta["T"]):
            assert np.array_equal(bits(obs), bits(g[U + "obs"][t])), (name, u, t)
            assert np.array_equal(done.astype(np.float32), g[U + "dones"][t])
            obs, rew, done = env.step(g[U + "actions"][t].reshape(meta["N"], -1)[:, 0].astype(np.int64))
            assert np.array_equal(rew, g[U + "rewards"][t])
        assert np.array_equal(bits(obs), bits(g[U + "next_obs"]))
        assert np.array_equal(done, g[U + "next_done"])
        st = env.episode_stats()
        ref = g[U + "ep_stats"]
        assert st["count"] == int(ref[2])
        if st["count"]:
            assert st["ep_len_mean"] == ref[0] and st["ep_rew_mean"] == ref[1]


@pytest.mark.parametrize("name", DISCRETE + MASKED)
def test_forward_teacher_forced(name):
    g, meta, hp, net = load(name)
    U = "u1/"
    B = meta["T"] * meta["N"]
    obs = g[U + "obs"].reshape(B, meta["obs"])
    acts = g[U + "actions"].reshape(B, -1)[:, :1].astype(np.int64)
    mask = g[U + "action_masks"].reshape(B, -1) if meta["masked"] else None
    lp, en, v = O.evaluate(net, g[U + "params_before"], obs, acts, mask)
    np.testing.assert_allclose(lp, g[U + "logprobs"].ravel(), rtol=0, atol=2e-6)   # tanh/exp/log ULP differences
    np.testing.assert_allclose(v, g[U + "values"].ravel(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(en, g[U + "rollout_entropy"].ravel(), rtol=0, atol=2e-6)
    nv = O.get_value(net, g[U + "params_before"], g[U + "next_obs"])
    np.testing.assert_allclose(nv, g[U + "next_value"].ravel(), rtol=0, atol=2e-6)


@pytest.mark.parametrize("name", DISCRETE + MASKED)
def test_minibatch_losses_grads_clip_adamw(name):
    g, meta, hp, net = load(name)
    U = "u1/"
    B = meta["T"] * meta["N"]
    MB = B // meta["nmb"]
    obs = g[U + "obs"].reshape(B, meta["obs"])
    acts = g[U + "actions"].reshape(B, -1) if meta["masked"] else g[U + "actions"].reshape(B)
    mask = g[U + "action_masks"].reshape(B, -1) if meta["masked"] else None
    perms = g[U + "perms"]
    scal = g[U + "step_scalars"]
    steps = scal.shape[0]
    for k in (0, 1, steps - 1):
        K = U + "k%d/" % k
        e, s = divmod(k, steps // meta["epochs"])
        idx = perms[e, s * MB:(s + 1) * MB]
        if k == 0:
            p_before = g[U + "params_before"]
            m0 = np.zeros_like(p_before)
            v0 = np.zeros_like(p_before)
        elif k == 1:
            p_before, m0, v0 = g[U + "k0/params_after"], g[U + "k0/exp_avg"], g[U + "k0/exp_avg_sq"]
        else:
            p_before = None  # parameters before the last step are not in the fixture: check step k via grads only below
        if p_before is not None:
            grads, st = O.minibatch_grads(net, hp, p_before, obs, acts, g[U + "logprobs"].ravel(), g[U + "gae_advantages"].ravel(),
                                          g[U + "gae_returns"].ravel(), g[U + "values"].ravel(), idx, mask)
            ref = dict(zip(O.STAT_NAMES + ("total_norm",), scal[k]))
            for n in O.STAT_NAMES:  # north_star: losses within 1e-5 fp32
                assert abs(st[n] - ref[n]) <= 1e-5 * max(1.0, abs(ref[n])), (name, k, n, st[n], ref[n])
            gref = g[K + "grads"]
            assert np.abs(grads - gref).max() <= 1e-6 + 1e-4 * np.abs(gref).max(), (name, k)
            clipped, total = O.clip_grad_norm(net, grads, hp.max_grad_norm)
            assert abs(total - ref["total_norm"]) <= 1e-5 * max(1.0, ref["total_norm"])
        # AdamW element-wise formula: bit-exact given the reference's own (clipped) gradients
        if p_before is not None:
            # clip with the reference's own recorded total_norm (clip_grad.h:76-80: float tensor arithmetic)
            total_ref = np.float32(scal[k, 6])
            coef = min(np.float32(hp.max_grad_norm) / (total_ref + np.float32(1e-6)), np.float32(1.0))
            cl_ref = (g[K + "grads"] * np.float32(coef)).astype(np.float32)
            lr = float(g[U + "lr"][0])
            p1, m1, v1 = O.adamw_step(p_before, cl_ref, m0, v0, lr, k + 1)
            assert np.array_equal(bits(m1), bits(g[K + "exp_avg"])), (name, k, "exp_avg")
            assert np.array_equal(bits(v1), bits(g[K + "exp_avg_sq"])), (name, k, "exp_avg_sq")
            # params: the formula is exact, but LibTorch's CPU sqrt goes through MKL VML (vsSqrt, not correctly
            # rounded): a handful of elements whose sqrt(v) sits ~0.49 ULP from a rounding boundary land 1 ULP away
            # in the quotient (a few ULP of the parameter after the add).
            ulp = np.abs(bits(p1).astype(np.int64) - bits(g[K + "params_after"]).astype(np.int64))
            assert ulp.max() <= 4 and (ulp != 0).mean() <= 2e-3, (name, k, int(ulp.max()), float((ulp != 0).mean()))


@pytest.mark.parametrize("name", DISCRETE)
def test_full_update_tracks_reference(name):
    """All epochs x minibatches of update 1 with the reference's permutations: parameters after the update
    stay within 2e-5 of the reference's (accumulated reduction-order noise over <= 40 optimizer steps)."""
    g, meta, hp, net = load(name)
    U = "u1/"
    B = meta["T"] * meta["N"]
    MB = B // meta["nmb"]
    obs = g[U + "obs"].reshape(B, meta["obs"])
    p = g[U + "params_before"].copy()
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    scal = g[U + "step_scalars"]
    k = 0
    for e in range(meta["epochs"]):
        for s in range(meta["nmb"]):
            idx = g[U + "perms"][e, s * MB:(s + 1) * MB]
            grads, st = O.minibatch_grads(net, hp, p, obs, g[U + "actions"].reshape(B), g[U + "logprobs"].ravel(),
                                          g[U + "gae_advantages"].ravel(), g[U + "gae_returns"].ravel(), g[U + "values"].ravel(), idx)
            for i, n in enumerate(O.STAT_NAMES):
                assert abs(st[n] - scal[k, i]) <= 2e-5 * max(1.0, abs(scal[k, i])), (name, k, n)
            grads, _ = O.clip_grad_norm(net, grads, hp.max_grad_norm)
            p, m, v = O.adamw_step(p, grads, m, v, float(g[U + "lr"][0]), k + 1)
            k += 1
    assert np.abs(p - g[U + "params_after"]).max() <= 2e-5
    ev = O.explained_variance(g[U + "gae_returns"], g[U + "values"])
    assert abs(ev - float(g[U + "explained_var"][0])) <= 1e-5


@pytest.mark.parametrize("name,envs,updates", [("curves_config0_8x128", 8, 150), ("curves_64x128", 64, 80)])
def test_reference_curve_fixtures(name, envs, updates):
    """tests/golden/curves_*.json (the unmodified reference's own table per update, ten seeds: oracle/make_curves.py) are what
    tests/test_gpu_curves.py compares the free-running build with: well-formed, complete, and the reference does learn in them
    (README.md:169-178: ep_len_mean climbs; every seed passes 195 and ends far above its start)."""
    import json
    with open(os.path.join(G, name + ".json")) as f:
        doc = json.load(f)
    cfg = doc["config"]
    assert cfg["num_envs"] == envs and cfg["num_steps"] == 128 and cfg["action_size"] == 2 and cfg["total_timesteps"] == updates * envs * 128
    assert [r["seed"] for r in doc["runs"]] == list(range(1, 11))
    for r in doc["runs"]:
        assert r["updates"] == updates and all(len(r[k]) == updates for k in ("ep_len_mean", "total_timesteps", "loss", "value_loss", "approx_kl"))
        assert r["total_timesteps"] == [envs * 128 * (u + 1) for u in range(updates)]
        el = [v for v in r["ep_len_mean"] if v is not None]
        assert max(el) >= 195 and el[-1] > 5 * el[1]
        assert r["loss"][0] is None and all(v is not None and np.isfinite(v) for v in r["loss"][1:])   # the first update's table has no train/ block


@pytest.mark.parametrize("name", ["headline_cartpole_4096x128", "headline_mountaincar_8192x128"])
def test_oracle_at_headline_size_against_the_compiled_reference(name):
    """The C restatement at BASELINE.json configs[1] / configs[3] FULL size against the compiled reference (oracle/ref_harness.cpp `headline`: the
    reference's components driven in train()'s order with hash-made actions; tests/test_gpu_headline_ref.py is the device's twin of this test):
    the whole rollout's env side (obs / rewards / dones / next_obs / next_done: 4096 x 128 and 8192 x 128 transitions, auto-resets included) and
    calcAdvantage on hash-made values -- CRC-32 for CRC-32; log-probs / values on the fixture's strided sample within fp32 noise."""
    import zlib
    g = O.read_pgld(os.path.join(G, name + ".pgld"))
    m, h = g["meta"], g["hparams"]
    T, N, obs_dim, A, max_steps, seed, masked = int(m[0]), int(m[1]), int(m[2]), int(m[3]), int(m[6]), int(m[7]), int(m[13])
    B = T * N
    crc = lambda a: zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF

    def mix64(x):
        with np.errstate(over="ignore"):
            x = x + np.uint64(0x9E3779B97F4A7C15)
            x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return x ^ (x >> np.uint64(31))
    unit24 = lambda hh: (hh >> np.uint64(40)).astype(np.float32) * np.float32(5.9604644775390625e-8)
    env = O.VecEnv(O.ENV_MOUNTAINCAR if masked else O.ENV_CARTPOLE, N, seed, max_steps)
    x = env.init()
    if masked:
        p0 = np.float32(-0.6) + np.float32(0.2) * unit24(mix64(np.uint64(0x4444 << 32) + np.arange(N, dtype=np.uint64)))
        x = np.stack([p0, np.zeros(N, np.float32)], 1).astype(np.float32)
        env.set_state(state=x, ep_len=np.zeros(N, np.int32), ep_rew=np.zeros(N, np.float32))
    assert crc(x) == int(g["crc_init_obs"][0])
    actions = (mix64(np.uint64(0x1111 << 32) + np.arange(B, dtype=np.uint64)) % np.uint64(A)).astype(np.int64).reshape(T, N)
    obs, rew, dones = np.empty((T, N, obs_dim), np.float32), np.empty((T, N), np.float32), np.empty((T, N), np.float32)
    done = np.zeros(N, np.float32)
    for t in range(T):
        obs[t], dones[t] = x, done
        x, r, d = env.step(actions[t])
        rew[t], done = r, d.astype(np.float32)
    assert crc(obs) == int(g["crc_obs"][0]) and crc(rew) == int(g["crc_rewards"][0]) and crc(dones) == int(g["crc_dones"][0])
    assert crc(x) == int(g["crc_next_obs"][0]) and crc(done.astype(np.int32)) == int(g["crc_next_done"][0])
    assert dones.sum() == g["count_done"][0]
    syn = (unit24(mix64(np.uint64(0x3333 << 32) + np.arange(B, dtype=np.uint64))) * np.float32(4.0) - np.float32(2.0)).reshape(T, N)
    adv, ret = O.gae(rew, syn, dones, np.full(N, 0.25, np.float32), done.astype(np.int32), float(h[1]), float(h[2]))
    assert crc(ret) == int(g["crc_syn_returns"][0]) and crc(adv) == int(g["crc_syn_advantages"][0])
    net = O.Net.make(obs_dim, [A], dist_kind=O.DIST_MASKED if masked else O.DIST_CATEGORICAL)
    rows = np.arange(0, B, 4099)
    mask = np.ones((rows.size, A), np.uint8) if masked else None
    lp, _, v = O.evaluate(net, g["params_before"], obs.reshape(B, obs_dim)[rows], actions.reshape(B)[rows], mask)
    np.testing.assert_allclose(lp, g["sample_logprobs"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(v, g["sample_values"], rtol=0, atol=3e-6)


# ---------------------------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[4]'s OWN SHAPE (obs 376, heads [3, 3, 3, 2], 4 x 256) against the compiled reference: oracle/ref_harness.cpp `config4` swaps
# 376 -> 256 x 4 -> {11 | 1} Sequentials into the unmodified Agent's public m_Actor / m_Critic (Agent.h:44-50) and re-drives PPO_MultiDiscrete's update
# on a hash-made batch (tests/c4_batch.py regenerates it; the fixture's CRCs hold that file to what the reference saw).  This pins the oracle's
# GENERIC mode (mlp_forward1 / mlp_backward1 at any width, depth and head list), which tests/test_gpu_generic.py uses as the yardstick for other shapes.
# ---------------------------------------------------------------------------------------------------------------------------------------
def _c4_load(name):
    import c4_batch as C4
    g = O.read_pgld(os.path.join(G, name + ".pgld"))
    m = C4.load_meta(g)
    params = O.read_pgld(os.path.join(G, "config4_params.pgld"))["params"]
    assert C4.crc(params) == int(g["crc_params_before"][0])
    net = O.Net.make(C4.O, list(C4.HEADS), hidden=m["hidden"], n_hidden=m["n_hidden"], dist_kind=O.DIST_MASKED)
    assert O.param_count(net) == params.size == 590860
    assert [tuple(s) for s in O.param_shapes(net)] == [tuple(int(v) for v in s) for s in g["param_shapes"]]     # critic first, then actor (Agent.cpp:65-66)
    hp = O.HParams(gamma=m["gamma"], gae_lambda=m["lam"], clip_coef=m["clip"], ent_coef=m["ent"], vf_coef=m["vf"], max_grad_norm=m["mgn"],
                   norm_adv=m["norm_adv"], clip_vloss=m["clip_vloss"])
    return C4, g, m, params.copy(), net, hp


def test_oracle_at_config4_shape_against_the_compiled_reference():
    """64 envs x 16 steps, 4 minibatches of 256 rows, 2 epochs, every per-sample tensor of the reference in full: forward (3e-6), calcAdvantage bit
    for bit, then the 8 optimizer steps -- losses 1e-5 (north_star; measured 9e-8), first and last gradient 5e-6 of its largest element (measured 7e-7),
    per-tensor gradient norms, parameters after the first step and after the update 2e-6 (measured 1.6e-7)."""
    C4, g, m, params, net, hp = _c4_load("config4_small_64x16")
    T, N = m["T"], m["N"]
    B, MB = T * N, T * N // m["nmb"]
    assert int(g["sample_stride"][0]) == 1
    b = C4.make_batch(T, N)
    C4.check_crcs(g, b)
    lp, en, v = O.evaluate(net, params, b["obs"], b["actions"], b["masks"])
    np.testing.assert_allclose(lp, g["sample_logprobs"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(en, g["sample_entropy"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(v, g["sample_values"], rtol=0, atol=3e-6)
    assert 0.5 < g["sample_entropy"].min() and g["sample_entropy"].max() <= 3 * np.log(3) + np.log(2) + 1e-5     # the masked distribution's TRUE entropy, summed over 4 heads
    np.testing.assert_allclose(O.get_value(net, params, b["next_obs"]), g["sample_next_value"], rtol=0, atol=3e-6)
    adv, ret = O.gae(b["rewards"].reshape(T, N), g["sample_values"].reshape(T, N), b["dones"].reshape(T, N), g["sample_next_value"], b["next_done"],
                     hp.gamma, hp.gae_lambda)
    assert np.array_equal(bits(adv).ravel(), bits(g["sample_advantages"])) and np.array_equal(bits(ret).ravel(), bits(g["sample_returns"]))
    shapes = [tuple(int(x) for x in s) for s in g["param_shapes"]]
    offs = np.cumsum([0] + [a * c for a, c in shapes])
    scal = g["step_scalars"]
    exp_avg, exp_avg_sq = np.zeros_like(params), np.zeros_like(params)
    k = 0
    for e in range(m["epochs"]):
        perm = C4.permutation(e, B)
        for s in range(m["nmb"]):
            gr, st = O.minibatch_grads(net, hp, params, b["obs"], b["actions"].astype(np.float32), g["sample_logprobs"], g["sample_advantages"], g["sample_returns"],
                                       g["sample_values"], perm[s * MB:(s + 1) * MB].astype(np.int64), b["masks"])
            for i, n in enumerate(O.STAT_NAMES):
                assert abs(st[n] - scal[k, i]) <= 1e-5 * max(1.0, abs(scal[k, i])), (k, n, st[n], scal[k, i])
            gc, total = O.clip_grad_norm(net, gr, hp.max_grad_norm)
            assert abs(total - scal[k, 6]) <= 2e-6 * scal[k, 6], (k, total, scal[k, 6])
            K = "k%d/" % k
            if K + "grad_norms" in g:
                norms = np.array([np.linalg.norm(gr[offs[i]:offs[i + 1]].astype(np.float64)) for i in range(len(shapes))])
                assert np.abs(norms - g[K + "grad_norms"]).max() <= 2e-6 * g[K + "grad_norms"].max(), k
                assert np.abs(gr[::61] - g[K + "sample_grads"]).max() <= 5e-6 * np.abs(gr).max(), k
            params, exp_avg, exp_avg_sq = O.adamw_step(params, gc, exp_avg, exp_avg_sq, float(g["lr"][0]), k + 1)
            if k == 0:
                assert np.abs(params[::61] - g["k0/sample_params_after"]).max() <= 2e-6
            k += 1
    assert k == scal.shape[0] == 8
    assert np.abs(params[::61] - g["sample_params_after"]).max() <= 2e-6
    assert g["max_abs_param_change"][0] > 1e-3


def test_oracle_on_the_config4_share_fixture():
    """The per-GPU share of configs[4] (2048 envs x 128 steps, minibatches of 65 536 rows, 40 steps): too large for the scalar oracle to follow the update, so here
    the hash-made batch against the reference's CRCs, the oracle's forward on the fixture's strided sample of rows, and the fixture's own sanity (the update
    the reference ran moved the policy: KL grows, value loss falls).  The device's twin of this fixture is tests/test_gpu_config4_ref.py."""
    C4, g, m, params, net, hp = _c4_load("config4_share_2048x128")
    T, N = m["T"], m["N"]
    assert (T, N, m["nmb"], m["epochs"]) == (128, 2048, 4, 10)
    b = C4.make_batch(T, N)
    C4.check_crcs(g, b)
    rows = np.arange(0, T * N, int(g["sample_stride"][0]))
    lp, en, v = O.evaluate(net, params, b["obs"][rows], b["actions"][rows], b["masks"][rows])
    np.testing.assert_allclose(lp, g["sample_logprobs"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(en, g["sample_entropy"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(v, g["sample_values"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(O.get_value(net, params, b["next_obs"]), g["sample_next_value"], rtol=0, atol=3e-6)
    scal = g["step_scalars"]
    assert scal.shape == (40, 7) and abs(scal[0, 3]) < 1e-8 and scal[0, 4] == 0.0           # first minibatch: ratio == 1, nothing clipped
    assert scal[-1, 3] > 5e-3 and scal[-1, 1] < scal[0, 1] and scal[-1, 0] < -1e-2         # the policy moved, the value loss fell
