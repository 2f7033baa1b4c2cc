#!/usr/bin/env python3
"""bench.py -- env-steps/sec (rollout + GAE + update) of the MI355X-native PPO hot path, BASELINE.json's metric.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W)

A "step" is one pass of the hot path over one batch: one iteration of PPO_Discrete::train()'s loop (reference
PPO/PPO_Discrete.cpp:511-659) = rollout of num_steps x num_envs env-steps, GAE scan, update_epochs x num_minibatches optimizer
steps.  Workload at N = 1: BASELINE.json configs[1] (CartPole-v1, 4096 envs x 128 steps, 2x64 MLP, 4 minibatches x 10 epochs,
hyper-parameters of the reference's CartPoleRecommendedSettings.toml with action_size = 2).  N > 1: every rank owns 4096 envs
(weak scaling, configs[2] at N = 8) and ONE RCCL all-reduce of the flat gradient per optimizer step crosses xGMI.
value = env-steps of all ranks / max-over-ranks wall time of the K timed steps; the reference prints the same quantity as `fps`
(PPO_Discrete.cpp:650-652,718).  Everything is resident in HBM when the timed region starts.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel (fused gather+forward+loss+backward): algorithmic FLOPs per launch / its average launch
                duration measured with HIP events on the kernel's own stream inside the timed region
  gae_roofline  the GAE scan (the kernel BASELINE.json's HBM-roofline target names), algorithmic bytes / event time
  cpu_baseline  the reference's own CPU ThreadPool path (oracle/_ref/ref_harness = the unmodified reference compiled against
                LibTorch CPU) timed on this host, or the C port when that binary is absent -- a reported baseline
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_PEAK_TFLOPS = 157.3   # MI355X dense f32 (vector = f32-input MFMA) peak, MI355X_MICROARCH.md "Chip-level parameters"
BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (same table)
HBM_PEAK_GBS = 8000.0     # HBM3E spec peak


def flops_per_sample(obs, act):
    """Algorithmic FLOPs of forward + backward of both MLPs for one sample (2 FLOP per MAC; backward = 2 x forward)."""
    macs = 0
    for out in (1, act):
        macs += obs * 64 + 64 * 64 + 64 * out
    return 3 * 2 * macs


def pmc_traffic(prefix):
    """HBM bytes per launch of the kernel whose name starts with `prefix`, from the newest committed rocprofv3 --pmc summary
    (profiles/*_pmc_per_dispatch.json, written by tools/collect_profiles.sh: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate
    passes).  Counters cannot be read from inside the timed run, so this is the committed measurement of the same command, or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_per_dispatch.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as fh:
            d = json.load(fh)
        for k, v in d.items():
            if k.startswith(prefix) and "hbm_read_bytes" in v and "hbm_write_bytes" in v:
                return {"bytes": v["hbm_read_bytes"] + v["hbm_write_bytes"], "read": v["hbm_read_bytes"], "write": v["hbm_write_bytes"],
                        "source": "profiles/" + os.path.basename(files[-1])}
    except Exception:
        return None
    return None


def cpu_baseline(num_envs, num_steps, obs, act):
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    cores = os.cpu_count() or 1
    if os.path.exists(ref):
        try:
            # bounded sample (~20 s on the GPU box's host cores): one full update iteration on a quarter of the envs; env-steps/s of the
            # reference's ThreadPool path is per-env-step work, so the rate carries to the full workload (measured 7.0k at 4096 envs x 2)
            updates = 1
            num_envs = min(num_envs, 1024)
            out = subprocess.run([ref, "bench", str(num_envs), str(num_steps), str(updates)], capture_output=True, text=True, timeout=900).stdout
            m = re.search(r"REF_BENCH (\{.*\})", out)
            if m:
                r = json.loads(m.group(1))
                return {"value": r["env_steps_per_sec"], "unit": "env-steps/s", "cores": int(r["threads"]), "kind": "reference",
                        "sample": "%d full update iteration(s) (rollout+GAE+update) of the unmodified reference's PPO_Discrete::train() on LibTorch "
                                  "CPU, ThreadPool(hardware_concurrency), %d envs x %d steps" % (updates, num_envs, num_steps)}
        except Exception as ex:  # fall through to the port
            sys.stderr.write("reference harness failed: %r\n" % (ex,))
    # C port (scalar, single thread): one rollout of a bounded env count + one minibatch of the update, scaled per env-step
    import numpy as np
    import oracle as O
    n = 256
    net = O.Net.make(obs, [act])
    rng = np.random.default_rng(0)
    params = (rng.standard_normal(O.param_count(net)) * 0.1).astype(np.float32)
    env = O.VecEnv(O.ENV_CARTPOLE, n, 2, 500)
    t0 = time.perf_counter()
    x = env.init()
    for t in range(num_steps):
        a, lp, en, v = O.act(net, params, x, 2, t)
        x, r, d = env.step(a[:, 0])
    t_roll = time.perf_counter() - t0
    B = n * num_steps
    hp = O.HParams(gamma=0.98, gae_lambda=0.95, clip_coef=0.2, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, norm_adv=1, clip_vloss=1)
    obs_b = rng.standard_normal((B, obs)).astype(np.float32)
    z = rng.standard_normal(B).astype(np.float32)
    t0 = time.perf_counter()
    O.minibatch_grads(net, hp, params, obs_b, (rng.random(B) < 0.5).astype(np.float32), z * 0.1 - 0.7, z, z, z, np.arange(B))
    t_upd = (time.perf_counter() - t0) * 10  # 10 epochs over the batch
    return {"value": B / (t_roll + t_upd), "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": "scalar C restatement, 1 thread: %d envs x %d steps rollout + 10 epochs of forward/backward over that batch" % (n, num_steps)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--num-steps", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--comm-selftest", action="store_true", help="N = 1 only: drive the multi-rank code path (RCCL all-reduces over a one-rank communicator, "
                    "three-kernel optimizer step) to see its per-step cost on one GPU; not a valid headline number")
    ap.add_argument("--profile", type=int, default=2, help="HIP-event timing inside the timed region: 0 off, 1 every kernel, 2 dominant kernel + GAE")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("--gpus %d needs one process per GPU: launch with `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus))
        args.gpus = world

    from __graft_entry__ import load_package
    P = load_package()
    dist = None
    if world > 1:
        dist, rank, world = P.dist.init_process_group("gloo")  # plumbing only: rendezvous, id broadcast, barrier, max over ranks
    obs, act = 4, 2
    N, T = args.envs, args.num_steps
    total_updates = args.steps + args.warmup
    cfg = P.dist.shard_config(P.make_config, rank, world, N * world, env_kind=P.ENV_CARTPOLE, dist_kind=P.DIST_CATEGORICAL, obs_size=obs,
                              head_dims=(act,), num_steps=T, num_minibatches=4, update_epochs=10, max_episode_steps=500, seed=2,
                              total_timesteps=total_updates * N * T * world, learning_rate=1e-3, gamma=0.98, gae_lambda=0.95, clip_coef=0.2,
                              ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, anneal_lr=True, device=local_rank)
    ctx = P.Context(cfg)
    if args.comm_selftest and world == 1:
        os.environ["PPO_COMM_SELFTEST"] = "1"
        ctx.comm_init(P.comm_unique_id(), 0, 1)
        del os.environ["PPO_COMM_SELFTEST"]
    P.dist.bootstrap_comm(ctx, dist, rank, world, P.comm_unique_id)
    ctx.init_orthogonal(2)   # same seed on every rank: replicated weights
    ctx.env_reset()

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()
        ctx.sync()

    for _ in range(args.warmup):
        ctx.train_iteration()
    ctx.profile_enable(args.profile)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.train_iteration()
    barrier()
    dt = time.perf_counter() - t0
    prof = ctx.profile_read()
    ctx.profile_enable(0)
    # GAE scan back to back on the context's own buffers, outside the timed region (same inputs -> same outputs): the kernel's time without
    # the ~3 us an event pair adds to a 5 us launch and without a foreign kernel in front of it (kernel-only time)
    class _Ptr:
        def __init__(self, p):
            self.ptr = p
    gb = [_Ptr(ctx.buffer_ptr(n)[0]) for n in ("REWARDS", "VALUES", "DONES", "NEXT_VALUE", "NEXT_DONE", "ADVANTAGES", "RETURNS")]
    reps = 200
    for _ in range(3):
        P.gae_launch(ctx, gb[0], gb[1], gb[2], gb[3], gb[4], args.num_steps, args.envs, 0.98, 0.95, gb[5], gb[6])
    ctx.sync()
    tg = time.perf_counter()
    for _ in range(reps):
        P.gae_launch(ctx, gb[0], gb[1], gb[2], gb[3], gb[4], args.num_steps, args.envs, 0.98, 0.95, gb[5], gb[6])
    ctx.sync()
    gae_b2b_ms = 1e3 * (time.perf_counter() - tg) / reps
    if dist is not None:
        dt = P.dist.max_over_ranks(dist, dt)
    st = ctx.stats()

    if rank == 0:
        env_steps = args.steps * N * T * world
        M = (N * T) // 4
        fl = flops_per_sample(obs, act) * M
        bf16_fl = 2 * ((M + 31) // 32) * 144 * (2 * 32 * 32 * 16)
        fb_ms = prof["fwd_bwd_ms"] / max(prof["fwd_bwd_launches"], 1) or float("nan")
        gae_ms = prof["gae_ms"] / max(prof["gae_launches"], 1) or float("nan")
        gae_bytes = 20 * N * T + 8 * N
        fb_tr, gae_tr = pmc_traffic("fwd_bwd_mfma_kernel"), pmc_traffic("gae_kernel")
        out = {
            "metric": "env-steps/sec (rollout+update)", "value": env_steps / dt, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic (fixed-seed CartPole-v1, random-init 2x64 actor/critic)",
            "config": {"workload": "CartPole-v1 PPO_Discrete, %d envs x %d steps per GPU, 2x64 MLP, 4 minibatches x 10 epochs (BASELINE.json configs[%d])"
                                   % (N, T, 1 if world == 1 else 2), "num_envs_per_gpu": N, "num_steps": T, "global_batch": N * T * world,
                       "minibatch_per_gpu": M, "optimizer_steps_per_step": 40, "parallelism": "dp%d (env-sharded, 1 RCCL grad all-reduce per optimizer step)" % world + (" [comm self-test]" if args.comm_selftest else "")},
            "roofline": {"kernel": "fwd_bwd_mfma_kernel (gather+forward+PPO loss+backward; fp32 via 3-term bf16 splits)", "bound": "mfma", "achieved": fl / (fb_ms * 1e-3) / 1e12,
                         "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fl / (fb_ms * 1e-3) / 1e12 / F32_PEAK_TFLOPS,
                         "traffic": (fb_tr or {}).get("bytes"), "traffic_detail": fb_tr,
                         "flops_per_launch": fl, "avg_launch_ms": fb_ms, "launches": prof["fwd_bwd_launches"],
                         "sampling": "HIP events around 1 launch in 8 (--profile 2), every launch with --profile 1",
                         # what the matrix cores actually execute: per 32-sample tile and net 144 v_mfma_f32_32x32x16_bf16 (fp32 products as
                         # six bf16 products over exact three-term splits) -- reported beside the algorithmic fp32 rate
                         "executed": {"unit": "TFLOP/s bf16", "achieved": bf16_fl / (fb_ms * 1e-3) / 1e12, "peak": BF16_PEAK_TFLOPS,
                                      "frac": bf16_fl / (fb_ms * 1e-3) / 1e12 / BF16_PEAK_TFLOPS}},
            "gae_roofline": {"kernel": "gae_kernel (exact mode)", "bound": "hbm", "achieved": gae_bytes / (gae_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": gae_bytes / (gae_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "traffic": (gae_tr or {}).get("bytes"), "traffic_detail": gae_tr, "bytes_per_launch": gae_bytes,
                             "avg_launch_ms": gae_ms, "launches": prof["gae_launches"],
                             "back_to_back": {"avg_launch_ms": gae_b2b_ms, "launches": reps, "achieved": gae_bytes / (gae_b2b_ms * 1e-3) / 1e9,
                                              "frac": gae_bytes / (gae_b2b_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                              "note": "same kernel, same buffers, 200 launches in a row after the timed region (wall time / 200): no event pair, no foreign kernel in front"}},
            "phase_ms_per_step": {"rollout": prof["rollout_ms"] / args.steps, "gae": prof["gae_ms"] / args.steps, "fwd_bwd": 40 * fb_ms,
                                  "grad_reduce": prof["reduce_ms"] / args.steps, "clip_adamw": prof["optimizer_ms"] / args.steps},
            "train_stats": {k: st[k] for k in ("loss", "ep_len_mean", "ep_rew_mean", "explained_variance", "global_step")},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, T, obs, act)
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
