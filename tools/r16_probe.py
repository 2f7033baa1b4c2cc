import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
P = load_package()
ctx = P.Context(P.make_config(num_envs=4096, num_steps=128, num_minibatches=4, update_epochs=1, seed=2, total_timesteps=4096 * 128 * 4))
ctx.init_orthogonal(2); ctx.env_reset()
for _ in range(2):
    ctx.rollout()
ctx.sync()
ctx.close()
