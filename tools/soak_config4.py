import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
P = load_package()
kw = dict(env_kind=P.ENV_SYNTHETIC, dist_kind=P.DIST_MASKED, obs_size=376, head_dims=(3, 3, 3, 2), hidden=256, n_hidden=4, num_envs=2048, num_steps=128,
          num_minibatches=4, update_epochs=10, max_episode_steps=200, seed=1, total_timesteps=60 * 2048 * 128, learning_rate=3e-4, gamma=0.99, gae_lambda=0.95, ent_coef=0.01,
          compute_dtype=P.DTYPE_BF16)
ctxs = [P.Context(P.make_config(**kw)) for _ in range(2)]
for c in ctxs:
    c.init_orthogonal(1); c.env_reset()
for it in range(40):
    for c in ctxs:
        c.train_iteration()
    if it % 10 == 9:
        p = [c.get_params() for c in ctxs]
        st = [c.stats() for c in ctxs]
        same = np.array_equal(p[0].view(np.uint32), p[1].view(np.uint32))
        print(it + 1, "bit-identical" if same else "DIFFERENT", st[0]["loss"], st[0]["ep_len_mean"], np.isfinite(p[0]).all(), flush=True)
        assert same
print("soak ok")
