"""Diagnostic: the gradient of ONE full-size minibatch (headline fixture, step 0) under the update kernel selected by DIAG_KERNEL (ws | mfma1 | valu -> ppo_config.kernel_flags), saved to
gpurun_out/grad_<kernel>.npy; with several files present prints per-tensor differences (of the tensor's largest element) against the vector kernel."""
import os, sys, glob, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_headline_ref as H
P = H.load_package()
g, meta = H.load("headline_cartpole_4096x128")
T, N, A = meta["T"], meta["N"], meta["act"]; B = T * N; MB = B // 4
KF = {"ws": 0, "mfma1": P.KERNEL_UPDATE_ONE_WAVE, "valu": P.KERNEL_UPDATE_VECTOR}[os.environ.get("DIAG_KERNEL", "ws")]
ctx = P.Context(P.make_config(num_envs=N, num_steps=T, num_minibatches=4, update_epochs=10, seed=2, total_timesteps=B, learning_rate=meta["lr"], gamma=meta["gamma"], gae_lambda=meta["lam"], kernel_flags=KF))
ctx.set_params(g["params_before"]); ctx.env_reset()
ctx.rollout((H.mix64(H.SEED_ACT + np.arange(B, dtype=np.uint64)) % np.uint64(A)).astype(np.int64).reshape(T, N, 1)); ctx.calc_advantage()
perm = np.argsort(H.mix64(H.SEED_PERM + np.arange(B, dtype=np.uint64)), kind="stable").astype(np.int32)
# a few optimizer steps first so that the policy is no longer uniform, then the gradient under test
for s in range(int(os.environ.get("DIAG_WARM", "0"))):
    ctx.minibatch_forward_backward(perm[(s % 4) * MB:(s % 4 + 1) * MB]); ctx.optimizer_step()
grads = ctx.minibatch_forward_backward(perm[:MB])
k = os.environ.get("DIAG_KERNEL", "ws")
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/grad_%s.npy" % k, grads); np.save("gpurun_out/params_%s.npy" % k, ctx.get_params())
names = ["c.W1", "c.b1", "c.W2", "c.b2", "c.W3", "c.b3", "a.W1", "a.b1", "a.W2", "a.b2", "a.W3", "a.b3"]
sizes = [256, 64, 4096, 64, 64, 1, 256, 64, 4096, 64, 128, 2]
if os.path.exists("gpurun_out/grad_valu.npy"):
    ref = np.load("gpurun_out/grad_valu.npy")
    for f in sorted(glob.glob("gpurun_out/grad_*.npy")):
        d = np.load(f); o = 0; out = []
        for n, sz in zip(names, sizes):
            a, b = d[o:o + sz], ref[o:o + sz]; out.append("%s=%.1e" % (n, np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))); o += sz
        print(os.path.basename(f), "params_same=%s" % np.array_equal(np.load(f.replace("grad_", "params_")), np.load("gpurun_out/params_valu.npy")), " ".join(out))
