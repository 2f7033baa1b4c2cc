"""MountainCar (masked categorical, returns down to -200) for 300 iterations at BASELINE configs[3] size: every statistic stays finite -- exercises the
fp16 range handling of the update kernel on value gradients two orders of magnitude larger than CartPole's.  python tools/soak_mountaincar.py"""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
P = load_package()
N, T, iters = 8192, 128, 300
ctx = P.Context(P.make_config(env_kind=P.ENV_MOUNTAINCAR, dist_kind=P.DIST_MASKED, obs_size=2, head_dims=(3,), num_envs=N, num_steps=T, num_minibatches=4, update_epochs=10,
                              max_episode_steps=200, seed=3, total_timesteps=iters * N * T, learning_rate=1e-3, gamma=0.99, gae_lambda=0.95, ent_coef=0.01, anneal_lr=True))
ctx.init_orthogonal(3); ctx.env_reset()
for i in range(iters):
    ctx.train_iteration()
    if (i + 1) % 50 == 0:
        st = ctx.stats()
        assert all(math.isfinite(st[k]) for k in ("loss", "pg_loss", "v_loss", "approx_kl", "total_norm", "ep_len_mean", "explained_variance")), st
        print(json.dumps({k: round(st[k], 5) for k in ("loss", "v_loss", "entropy_loss", "approx_kl", "total_norm", "ep_len_mean", "ep_rew_mean", "explained_variance")}), flush=True)
ctx.close()
print("mountaincar soak ok")
