"""Diagnostic: distance of the 40 optimizer steps of the headline fixture from the compiled reference, per update kernel (DIAG_KERNEL = ws | mfma1 | valu)."""
import os, sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_headline_ref as H
import oracle as O
P = H.load_package()
name = "headline_cartpole_4096x128"
g, meta = H.load(name)
T, N, O_, A = meta["T"], meta["N"], meta["obs"], meta["act"]; B = T*N; MB = B//meta["nmb"]
KF = {"ws": 0, "mfma1": P.KERNEL_UPDATE_ONE_WAVE, "valu": P.KERNEL_UPDATE_VECTOR}[os.environ.get("DIAG_KERNEL", "ws")]   # which update kernel: ppo_config.kernel_flags
ctx = P.Context(P.make_config(num_envs=N, num_steps=T, num_minibatches=4, update_epochs=10, seed=2, total_timesteps=B, learning_rate=meta["lr"], gamma=meta["gamma"], gae_lambda=meta["lam"], kernel_flags=KF))
ctx.set_params(g["params_before"]); ctx.env_reset()
actions = (H.mix64(H.SEED_ACT + np.arange(B, dtype=np.uint64)) % np.uint64(A)).astype(np.int64).reshape(T, N, 1)
ctx.rollout(actions); ctx.calc_advantage(); ctx.set_learning_rate(float(g["lr"][0]))
scal = g["step_scalars"]; k = 0
names = O.STAT_NAMES + ("total_norm",)
for e in range(10):
    keys = H.mix64(H.SEED_PERM + np.uint64(e * B) + np.arange(B, dtype=np.uint64))
    perm = np.argsort(keys, kind="stable").astype(np.int32)
    for s in range(4):
        ctx.minibatch_forward_backward(perm[s*MB:(s+1)*MB]); ctx.optimizer_step(); st = ctx.stats()
        if k in (0, 1, 9, 19, 29, 38, 39):
            print(k, " ".join("%s=%.2e" % (n, abs(st[kk] - scal[k][i])) for i, (n, kk) in enumerate(zip(names, ("pg_loss","v_loss","entropy_loss","approx_kl","clipfrac_last","loss","total_norm")))), "ref_norm=%.4f" % scal[k][6])
        k += 1
print("params_after maxdiff %.3e" % np.abs(ctx.get_params() - g["params_after"]).max())
