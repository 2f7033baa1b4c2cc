"""BASELINE.json configs[4]'s OWN SHAPE -- obs 376, heads [3, 3, 3, 2], 4 x 256 tanh bodies -- against the COMPILED REFERENCE.

The reference's Agent hard-wires 2 x 64 bodies (Agent.cpp:25-59), but m_Critic / m_Actor / m_actionSpace / m_actionSpaceSum are public members
(Agent.h:44-50) and getActionAndValueMasked / getValue (Agent.cpp:107-170) are shape-agnostic: `oracle/ref_harness.cpp config4` swaps 376 -> 256 x 4 ->
{11 | 1} Sequentials (initialised by the reference's own ppoLayerInit) into an unmodified PPO_MultiDiscrete's agent and re-drives its update
(PPO_MultiDiscrete.cpp:593-668: the harness's certified minibatch expressions, clip_grad_norm_, AdamW) on a batch whose every input is a counter hash
(tests/c4_batch.py regenerates it, held to the reference's CRCs).  Fixtures: tests/golden/config4_params.pgld (the reference's initial parameters in full),
config4_small_64x16.pgld (every per-sample tensor in full), config4_share_2048x128.pgld (configs[4]'s per-GPU share: 16 384 envs / 8 GPUs, minibatches
of 65 536 rows, 40 optimizer steps: strided samples + binary64 sums + the 7 scalars of every step).

The f32 generic path (compute_dtype 0) is held to the bars of the 2 x 64 tests: forward 5e-6 (values of magnitude ~2 from K = 376 sums), advantages 2e-5, losses
1e-5 (north_star) on every optimizer step -- with ONE qualification that the reference itself supplies.  PPO's loss has kinks (value clipping, the ratio clip,
max(unclipped, clipped)): once the first samples of a minibatch reach a kink, two CORRECT fp32 trajectories a rounding apart separate visibly (one sample changing
branch moves the gradient by 1 / M of its own gradient).  The fixture therefore carries a TWIN: the same scenario run by the same unmodified reference from
parameters that differ from the fixture's in the last bit of every element (`ulp_twin/*`).  Over the share's 40 steps the twin stays within 3e-7 of the fixture
for 23 steps and then drifts to 3e-5 (value loss), 4e-4 (gradient norm), 4e-4 (parameters); the device is held, per step, to max(1e-5, 4 x the twin's
distance so far) -- measured: 1e-7 for 29 steps, then 1.5e-5 / 3.9e-4 / 2e-4, i.e. INSIDE the reference's own sensitivity.  The bf16 path (compute_dtype 1: "4x256 MLP bf16 with MFMA GEMMs", BASELINE configs[4]) has no reference
counterpart; its distance from the reference's fp32 numbers is measured and printed beside it, with a bar at ~3x of it -- a distance, not a parity claim.
"""
import os

import numpy as np
import pytest

import c4_batch as C4
import oracle as O
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def P():
    return load_package()


@pytest.fixture(scope="module")
def share_batch():
    return C4.make_batch(128, 2048)


def _ctx(P, m, dtype):
    return P.Context(P.make_config(env_kind=P.ENV_SYNTHETIC, dist_kind=P.DIST_MASKED, obs_size=C4.O, head_dims=C4.HEADS, hidden=m["hidden"], n_hidden=m["n_hidden"],
                                   num_envs=m["N"], num_steps=m["T"], num_minibatches=m["nmb"], update_epochs=m["epochs"], max_episode_steps=m["max_steps"],
                                   use_gae=True, norm_adv=bool(m["norm_adv"]), clip_vloss=bool(m["clip_vloss"]), anneal_lr=bool(m["anneal"]), seed=m["seed"],
                                   total_timesteps=m["T"] * m["N"], learning_rate=m["lr"], gamma=m["gamma"], gae_lambda=m["lam"], clip_coef=m["clip"],
                                   ent_coef=m["ent"], vf_coef=m["vf"], max_grad_norm=m["mgn"], compute_dtype=dtype))


def _drive(P, name, batch, dtype):
    """Writes the hash-made batch into the context's rollout buffers, lets the device compute what the reference's rollout computes (log-probs, values, the
    bootstrap value, advantages), runs the update with the injected permutations.  Returns the worst distance from the reference per quantity."""
    g = O.read_pgld(os.path.join(G, name + ".pgld"))
    m = C4.load_meta(g)
    params0 = O.read_pgld(os.path.join(G, "config4_params.pgld"))["params"]
    assert C4.crc(params0) == int(g["crc_params_before"][0])
    T, N = m["T"], m["N"]
    B, MB = T * N, T * N // m["nmb"]
    b = batch if batch is not None else C4.make_batch(T, N)
    C4.check_crcs(g, b)
    stride, pstride = int(g["sample_stride"][0]), int(g["sample_stride"][1])
    ctx = _ctx(P, m, dtype)
    assert ctx.P == params0.size
    assert [tuple(s) for s in ctx.param_shapes()] == [tuple(int(v) for v in s) for s in g["param_shapes"]]
    ctx.set_params(params0)
    ctx.write("OBS", b["obs"])
    ctx.write("MASKS", b["masks"])
    ctx.write("ACTIONS", b["actions"].astype(np.int32))
    ctx.write("REWARDS", b["rewards"])
    ctx.write("DONES", b["dones"])
    ctx.write("NEXT_OBS", b["next_obs"])
    ctx.write("NEXT_DONE", b["next_done"])
    d = {}
    # ---- Agent::getActionAndValueMasked on the whole batch, teacher-forced (Agent.cpp:137-170) ----
    lp, en, v = np.empty(B, np.float32), np.empty(B, np.float32), np.empty(B, np.float32)
    for lo in range(0, B, 65536):
        hi = min(B, lo + 65536)
        a, lp[lo:hi], en[lo:hi], v[lo:hi] = ctx.policy_act(b["obs"][lo:hi], mask=b["masks"][lo:hi], action=b["actions"][lo:hi])
        assert np.array_equal(a, b["actions"][lo:hi])
    for key, arr in (("logprobs", lp), ("entropy", en), ("values", v)):
        d[key] = float(np.abs(arr[::stride] - g["sample_" + key]).max())
        s = arr.astype(np.float64)
        d["sum_" + key] = abs(s.sum() - g["sums_" + key][0]) / B
    ctx.write("LOGPROBS", lp)
    ctx.write("VALUES", v)
    adv, ret = ctx.calc_advantage()
    d["next_value"] = float(np.abs(ctx.read("NEXT_VALUE") - g["sample_next_value"]).max())
    d["advantages"] = float(np.abs(adv.reshape(-1)[::stride] - g["sample_advantages"]).max())
    d["returns"] = float(np.abs(ret.reshape(-1)[::stride] - g["sample_returns"]).max())
    # the scan itself, bit for bit, on the device's own values (the reference's at this shape differ by fp32 noise)
    adv_o, ret_o = O.gae(b["rewards"].reshape(T, N), v.reshape(T, N), b["dones"].reshape(T, N), ctx.read("NEXT_VALUE"), b["next_done"], m["gamma"], m["lam"])
    assert np.array_equal(adv.view(np.uint32), adv_o.view(np.uint32)) and np.array_equal(ret.view(np.uint32), ret_o.view(np.uint32))
    # ---- the update (PPO_MultiDiscrete.cpp:593-668) ----
    ctx.set_learning_rate(float(g["lr"][0]))
    scal = g["step_scalars"]
    shapes = [tuple(int(x) for x in s) for s in g["param_shapes"]]
    offs = np.cumsum([0] + [a_ * c for a_, c in shapes])
    worst = {n: 0.0 for n in O.STAT_NAMES + ("total_norm",)}
    names = O.STAT_NAMES + ("total_norm",)
    # the reference's own sensitivity: distance of its one-ulp twin from the fixture, per step, as a running maximum (a trajectory that has left does not come back)
    twin = g["ulp_twin/step_scalars"]
    twin_d = np.abs(twin - scal) / np.where(np.arange(7) == 4, 1.0, np.maximum(1.0, np.abs(scal)))
    twin_env = np.maximum.accumulate(twin_d, axis=0)
    excess = {n: 0.0 for n in names}     # worst of (device distance / per-step bar); <= 1 passes
    trace = []
    k = 0
    for e in range(m["epochs"]):
        perm = C4.permutation(e, B)
        for s in range(m["nmb"]):
            grads = ctx.minibatch_forward_backward(perm[s * MB:(s + 1) * MB])
            K = "k%d/" % k
            if K + "grad_norms" in g:
                norms = np.array([np.linalg.norm(grads[offs[i]:offs[i + 1]].astype(np.float64)) for i in range(len(shapes))])
                d[K + "grad_norms"] = float(np.abs(norms - g[K + "grad_norms"]).max() / g[K + "grad_norms"].max())
                d[K + "grads"] = float(np.abs(grads[::pstride] - g[K + "sample_grads"]).max() / np.abs(g[K + "sample_grads"]).max())
            ctx.optimizer_step()
            st = ctx.stats()
            for i, (n, key) in enumerate((("pg_loss", "pg_loss"), ("v_loss", "v_loss"), ("entropy_loss", "entropy_loss"), ("approx_kl", "approx_kl"),
                                          ("clipfrac", "clipfrac_last"), ("loss", "loss"), ("total_norm", "total_norm"))):
                dist = abs(st[key] - scal[k, i]) / (1.0 if n == "clipfrac" else max(1.0, abs(scal[k, i])))
                worst[n] = max(worst[n], dist)
                floor = {"clipfrac": 1e-4, "total_norm": 2e-6}.get(n, 1e-5)
                excess[n] = max(excess[n], dist / max(floor, 4.0 * twin_env[k, i]))
                if n in ("v_loss", "total_norm"):
                    trace.append(dist)
            if k == 0:
                d["k0/params_after"] = float(np.abs(ctx.get_params()[::pstride] - g["k0/sample_params_after"]).max())
            k += 1
    assert k == scal.shape[0]
    print("per step |v_loss - ref|, |total_norm - ref| (relative):", " ".join("%.0e/%.0e" % (trace[2 * i], trace[2 * i + 1]) for i in range(k)))
    p_after = ctx.get_params()
    d["params_after"] = float(np.abs(p_after[::pstride] - g["sample_params_after"]).max())
    norms = np.array([np.linalg.norm(p_after[offs[i]:offs[i + 1]].astype(np.float64)) for i in range(len(shapes))])
    d["norms_params_after"] = float(np.abs(norms - g["norms_params_after"]).max() / g["norms_params_after"].max())
    assert np.abs(p_after - params0).max() > 1e-3
    ctx.close()
    d.update({"step/" + n: w for n, w in worst.items()})
    d.update({"excess/" + n: w for n, w in excess.items()})
    d["twin/params_after"] = float(np.abs(g["ulp_twin/sample_params_after"] - g["sample_params_after"]).max())
    last = "k%d/sample_grads" % (k - 1)
    d["twin/last_grads"] = float(np.abs(g["ulp_twin/sample_last_grads"] - g[last]).max() / np.abs(g[last]).max())
    d["twin/steps"] = {n: float(twin_env[-1, i]) for i, n in enumerate(names)}
    return d, m


def _report(name, dtype, d):
    print("%s %s against the compiled reference:" % (name, ("f32", "bf16")[dtype]), {k: ("%.2e" % x if not isinstance(x, dict) else {a: "%.1e" % b for a, b in x.items()}) for k, x in d.items()})


# bars for the f32 path: the 2 x 64 tests' (tests/test_gpu_generic.py TOL[0], tests/test_gpu_headline_ref.py).  The per-step scalars are held through
# "excess/*" = distance / max(floor, 4 x the reference twin's distance so far), floor = 1e-5 (north_star; clipfrac 1e-4: a count; total_norm 2e-6)
F32_BARS = {"logprobs": 5e-6, "entropy": 5e-6, "values": 5e-6, "next_value": 5e-6, "sum_logprobs": 1e-6, "sum_entropy": 1e-6, "sum_values": 1e-6,
            "advantages": 2e-5, "returns": 2e-5, "excess/pg_loss": 1.0, "excess/v_loss": 1.0, "excess/entropy_loss": 1.0, "excess/approx_kl": 1.0, "excess/loss": 1.0,
            "excess/clipfrac": 1.0, "excess/total_norm": 1.0, "k0/grad_norms": 1e-5, "k0/grads": 2e-5, "k0/params_after": 2e-6}


def _hold(d, bars, name):
    for key, bar in bars.items():
        assert d[key] <= bar, (name, key, d[key], bar)
    # end of the update: parameters and the last gradient within 2e-6 / 2e-5, or 4 x what the reference's own twin shows
    assert d["params_after"] <= max(2e-6, 4.0 * d["twin/params_after"]), (name, d["params_after"], d["twin/params_after"])
    last = [k for k in d if k.startswith("k") and k.endswith("/grads") and k != "k0/grads"]
    for k in last:
        assert d[k] <= max(2e-5, 4.0 * d["twin/last_grads"]), (name, k, d[k], d["twin/last_grads"])


def test_f32_path_at_config4_shape_small_against_the_compiled_reference(P):
    d, m = _drive(P, "config4_small_64x16", None, 0)
    _report("config4_small_64x16", 0, d)
    _hold(d, F32_BARS, "small")
    assert d["step/v_loss"] <= 1e-5 and d["step/loss"] <= 1e-5 and d["params_after"] <= 2e-6    # 8 steps: no sample reaches a kink, plain 1e-5 / 2e-6 hold


def test_f32_path_at_config4_share_against_the_compiled_reference(P, share_batch):
    """2048 envs x 128 steps, minibatches of 65 536 rows, all 40 optimizer steps of the reference's update."""
    d, m = _drive(P, "config4_share_2048x128", share_batch, 0)
    _report("config4_share_2048x128", 0, d)
    _hold(d, F32_BARS, "share")


def test_bf16_path_at_config4_share_distance_from_the_compiled_reference(P, share_batch):
    """The arithmetic BASELINE configs[4] names (bf16 operands and activations, f32 accumulation and master weights) on the same batch: how far it sits from
    the reference's fp32 numbers over the 40 steps.  A measured distance with regression FENCES at ~3x of it -- no parity claim; the parity claim is the f32 test above, and past the
    first ~29 steps that one is itself relaxed from 1e-5 to 4x the reference's own one-ulp-twin distance."""
    d, m = _drive(P, "config4_share_2048x128", share_batch, 1)
    _report("config4_share_2048x128", 1, d)
    assert all(np.isfinite(x) for x in d.values() if not isinstance(x, dict))
    # measured (round 5): log-prob 1.6e-4, value 7.9e-3, advantage 8.6e-3, pg 2.7e-5, value loss 2.2e-4, entropy 7.1e-6, KL 1.8e-5, loss 1.5e-4, parameters 2.1e-3
    bars = {"logprobs": 5e-4, "entropy": 1e-5, "values": 2.5e-2, "advantages": 3e-2, "step/pg_loss": 1e-4, "step/v_loss": 7e-4, "step/entropy_loss": 3e-5,
            "step/approx_kl": 6e-5, "step/loss": 5e-4, "params_after": 7e-3}
    for key, bar in bars.items():
        assert d[key] <= bar, ("share bf16", key, d[key], bar)
