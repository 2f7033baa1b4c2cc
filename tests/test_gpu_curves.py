"""Learning-curve parity (north_star: "loss/return curves matching the reference"; BASELINE.md section 4: "learning curves (ep_len_mean vs
steps) matching within seed noise").

The reference's own acceptance test is "run ./PPO and watch ep_len_mean" (README.md:169-178; the table of PPO_Discrete.cpp:700-774).
tests/golden/curves_*.json hold that table, per update, for ten seeds of the UNMODIFIED `PPO_Discrete::train()` (oracle/ref_harness.cpp `curves`,
oracle/make_curves.py) at BASELINE.json configs[0] (8 envs x 128 steps) and at 64 x 128, CartPoleRecommendedSettings.toml's hyper-parameters with
action_size = 2.  Sampling (torch's CPU mt19937) and the orthogonal init (LAPACK) cannot be reproduced bit for bit, so the
free-running build is compared with the reference as two SAMPLES of seeds:

  * steps-to-solve  = the first `total_timesteps` at which ep_len_mean (CircularBuffer(100), Utils.h:30-79) reaches 195
  * plateau         = the mean of ep_len_mean over the last 10 % of the updates

Band rule, per statistic: (a) the build's median lies inside the reference's [min, max] over its seeds, and (b) a two-sided Mann-Whitney U test
does not separate the two samples at p = 0.01.  The curves' pointwise medians are also compared over the climb (the build's median curve stays
inside the reference's seed envelope widened by 15 % for at least 90 % of the updates).

The LOSS half of the table (PPO_Discrete.cpp:747-771: value_loss, explained_variance, clip_fraction, approx_kl, policy_gradient_loss, loss -- last-minibatch
values of the update, clip_fraction the mean over its minibatches) is held to the same band rule per update: the build's median over its seeds inside the
reference's seed envelope (widened by 15 % of its width + a floor for quantities that go to zero) on at least 90 % of the updates, and Mann-Whitney on the
area under value_loss.  `curves_4096x128` is BASELINE.json configs[1]'s own shape (4096 envs x 128 steps, 30 updates, five seeds of the reference).
"""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


SEEDS = list(range(1, 11))
SOLVED = 195.0
# the table's row name (printPPOResults, PPO_Discrete.cpp:747-771) -> (field of ppo_stats, floor of the envelope's widening)
LOSS_KEYS = {"value_loss": ("v_loss", 1e-3), "explained_variance": ("explained_variance", 1e-3), "clip_fraction": ("clipfrac_mean", 2e-3),
             "approx_kl": ("approx_kl", 2e-4), "policy_gradient_loss": ("pg_loss", 5e-4), "loss": ("loss", 1e-3)}


@pytest.fixture(scope="module")
def P():
    return load_package()


def load_curves(name):
    with open(os.path.join(HERE, "golden", name + ".json")) as f:
        return json.load(f)


def steps_to_solve(steps, ep_len):
    for s, v in zip(steps, ep_len):
        if v is not None and v >= SOLVED:
            return float(s)
    return float("inf")


def plateau(ep_len):
    k = max(1, len(ep_len) // 10)
    return float(np.mean([v for v in ep_len[-k:] if v is not None]))


def run_build(P, cfg, seed):
    c = P.make_config(num_envs=cfg["num_envs"], num_steps=cfg["num_steps"], num_minibatches=cfg["num_minibatches"], update_epochs=cfg["update_epochs"],
                      seed=seed, total_timesteps=cfg["total_timesteps"], learning_rate=cfg["learning_rate"], gamma=cfg["gamma"],
                      gae_lambda=cfg["gae_lambda"], clip_coef=cfg["clip_coef"], ent_coef=cfg["ent_coef"], vf_coef=cfg["vf_coef"],
                      max_grad_norm=cfg["max_grad_norm"], max_episode_steps=cfg["max_episode_steps"], norm_adv=cfg["norm_adv"],
                      clip_vloss=cfg["clip_vloss"], anneal_lr=cfg["anneal_lr"])
    ctx = P.Context(c)
    ctx.init_orthogonal(seed)
    ctx.env_reset()
    updates = cfg["total_timesteps"] // (cfg["num_envs"] * cfg["num_steps"])
    steps, ep_len = [], []
    table = {k: [] for k in LOSS_KEYS}
    for _ in range(updates):
        ctx.train_iteration()
        st = ctx.stats()
        steps.append(st["global_step"])
        ep_len.append(st["ep_len_mean"] if st["ep_count"] > 0 else None)
        for k, (mine, _) in LOSS_KEYS.items():
            table[k].append(st[mine])
    ctx.close()
    assert all(np.all(np.isfinite(v)) for v in table.values())
    return steps, ep_len, table


def summarize(P, name):
    from scipy.stats import mannwhitneyu

    doc = load_curves(name)
    cfg = doc["config"]
    ref_solve = [steps_to_solve(r["total_timesteps"], r["ep_len_mean"]) for r in doc["runs"]]
    ref_plat = [plateau(r["ep_len_mean"]) for r in doc["runs"]]
    runs = [run_build(P, cfg, s) for s in SEEDS]
    our_solve = [steps_to_solve(st, el) for st, el, _ in runs]
    our_plat = [plateau(el) for _, el, _ in runs]
    out = {"scenario": name, "ref_steps_to_195": ref_solve, "build_steps_to_195": our_solve, "ref_plateau": ref_plat, "build_plateau": our_plat}
    for key, ref, ours in (("steps_to_195", ref_solve, our_solve), ("plateau", ref_plat, our_plat)):
        med = float(np.median(ours))
        p = float(mannwhitneyu(ours, ref, alternative="two-sided").pvalue)
        out[key] = {"build_median": med, "ref_min": float(min(ref)), "ref_median": float(np.median(ref)), "ref_max": float(max(ref)), "mannwhitney_p": p}
    # pointwise: the build's median curve against the reference's seed envelope (update 1's table prints ep_len_mean with one digit: skipped)
    U = min(len(runs[0][1]), min(r["updates"] for r in doc["runs"]))
    ref = np.array([[np.nan if v is None else v for v in r["ep_len_mean"][:U]] for r in doc["runs"]], dtype=np.float64)
    ours = np.array([[np.nan if v is None else v for v in el[:U]] for _, el, _ in runs], dtype=np.float64)
    lo, hi = np.nanmin(ref[:, 1:], axis=0), np.nanmax(ref[:, 1:], axis=0)
    med = np.nanmedian(ours[:, 1:], axis=0)
    inside = (med >= 0.85 * lo) & (med <= 1.15 * hi)
    out["median_curve_inside_envelope"] = float(np.mean(inside))
    # the loss half of the table, per update (the first update's table has no train/ block: the reference's row is null there)
    out["losses"] = {}
    for key, (_, floor) in LOSS_KEYS.items():
        ref = np.array([[np.nan if v is None else v for v in r[key][:U]] for r in doc["runs"]], dtype=np.float64)[:, 1:]
        ours = np.array([t[key][:U] for _, _, t in runs], dtype=np.float64)[:, 1:]
        lo, hi = np.nanmin(ref, axis=0), np.nanmax(ref, axis=0)
        pad = 0.15 * (hi - lo) + floor
        med = np.median(ours, axis=0)
        inside = (med >= lo - pad) & (med <= hi + pad)
        area_ref, area_ours = np.nansum(ref, axis=1), np.sum(ours, axis=1)
        out["losses"][key] = {"inside_envelope": float(np.mean(inside)), "build_area_median": float(np.median(area_ours)), "ref_area_min": float(area_ref.min()),
                              "ref_area_median": float(np.median(area_ref)), "ref_area_max": float(area_ref.max()),
                              "area_mannwhitney_p": float(mannwhitneyu(area_ours, area_ref, alternative="two-sided").pvalue),
                              "build_median_last": float(med[-1]), "ref_median_last": float(np.nanmedian(ref[:, -1]))}
    return out


SCENARIOS = ["curves_config0_8x128", "curves_64x128", "curves_4096x128"]


@pytest.mark.parametrize("name", SCENARIOS)
def test_learning_curves_match_reference_band(P, name):
    r = summarize(P, name)
    print(json.dumps(r))
    for key in ("steps_to_195", "plateau"):
        s = r[key]
        assert s["ref_min"] <= s["build_median"] <= s["ref_max"], (name, key, r)
        assert s["mannwhitney_p"] > 0.01, (name, key, r)
    assert r["median_curve_inside_envelope"] >= 0.90, (name, r)
    assert all(np.isfinite(v) for v in r["build_steps_to_195"]), (name, "a seed of the build never reached ep_len_mean 195", r)
    # the loss curves (north_star: "loss/return curves matching the reference")
    for key, s in r["losses"].items():
        assert s["inside_envelope"] >= 0.90, (name, key, s)
    assert r["losses"]["value_loss"]["area_mannwhitney_p"] > 0.01, (name, r["losses"]["value_loss"])


if __name__ == "__main__":   # python tests/test_gpu_curves.py -> the numbers the test compares (profiles/r05_*_curves.json)
    P_ = load_package()
    print(json.dumps([summarize(P_, n) for n in SCENARIOS], indent=1))
