#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Generates tests/golden/curves_*.json: the learning curves of the UNMODIFIED reference.

For every seed `oracle/_ref/ref_harness curves` runs the reference's own `PPO_Discrete::train()` (PPO/PPO_Discrete.cpp:485-690) with
CartPoleRecommendedSettings.toml's hyper-parameters and `action_size = 2`, and parses the table it prints per update
(printPPOResults, :700-774) -- the reference's de-facto acceptance test (README.md:169-178).  Runs only in the build container
(needs /root/reference compiled into oracle/_ref); the JSON travels, the reference does not.

    python oracle/make_curves.py [name ...]   # writes tests/golden/<name>.json for every (or the named) scenario below
"""
import concurrent.futures as cf
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")

# name -> (num_envs, num_steps, total_timesteps, seeds).  total_timesteps fixes the LR-anneal horizon (PPO_Discrete.cpp:496, 514-518),
# so it is part of the scenario; the GPU test runs the build with the same value.
SCENARIOS = {
    "curves_config0_8x128": (8, 128, 150 * 8 * 128, list(range(1, 11))),     # BASELINE.json configs[0] shape, 150 updates
    "curves_64x128": (64, 128, 80 * 64 * 128, list(range(1, 11))),           # 80 updates of 8192 steps
    "curves_4096x128": (4096, 128, 30 * 4096 * 128, list(range(1, 11))),      # BASELINE.json configs[1] shape: 30 updates of 524 288 steps (~10 s each here)
}


def one(num_envs, num_steps, total, seed):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "c.json")
        subprocess.check_call([HARNESS, "curves", out, str(num_envs), str(num_steps), str(total), str(seed)], cwd=d,
                              stdout=subprocess.DEVNULL)
        return json.load(open(out))


def main():
    if not os.path.exists(HARNESS):
        sys.exit("oracle/_ref/ref_harness missing: run `make -C oracle ref` in the build container")
    wanted = sys.argv[1:] or list(SCENARIOS)
    for name in wanted:
        n, t, total, seeds = SCENARIOS[name]
        with cf.ThreadPoolExecutor(max_workers=1) as ex:
            runs = list(ex.map(lambda s: one(n, t, total, s), seeds))
        doc = {
            "source": "unmodified reference PPO_Discrete::train() on LibTorch CPU, table of printPPOResults parsed by oracle/ref_harness.cpp curves",
            "config": {"num_envs": n, "num_steps": t, "total_timesteps": total, "action_size": 2, "max_episode_steps": 500, "learning_rate": 0.001,
                       "anneal_lr": True, "gamma": 0.98, "gae_lambda": 0.95, "num_minibatches": 4, "update_epochs": 10, "clip_coef": 0.2,
                       "ent_coef": 0.0, "vf_coef": 0.5, "max_grad_norm": 0.5, "norm_adv": True, "clip_vloss": True},
            "runs": runs,
        }
        path = os.path.join(ROOT, "tests", "golden", name + ".json")
        with open(path, "w") as f:
            json.dump(doc, f, separators=(",", ":"))
        print("wrote", path, os.path.getsize(path), "bytes;", "final ep_len_mean per seed:",
              [r["ep_len_mean"][-1] for r in runs])


if __name__ == "__main__":
    main()
