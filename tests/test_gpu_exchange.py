"""The one-shot direct exchange (ppo_hip.h: ppo_comm_exchange_handle / ppo_comm_init_exchange) with REAL separate processes: 2 and 4 ranks share
the one GPU of the test box (each its own HIP context, the exchange buffers mapped through HIP IPC), bootstrap over torch.distributed (gloo) as
bench.py does.  Every rank trains its env shard for two iterations; the replicas must end bit-identical to each other (the exchange sums in rank
order everywhere) and equal, to float noise, to ONE context that owns all the envs -- the equivalence contract of SURVEY.md 8(e).
What this cannot show is the xGMI hop: on the box the peers' memory is the same device's."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

WORKER = r"""
import json, os, sys
sys.path.insert(0, {root!r})
import numpy as np
from __graft_entry__ import load_package
P = load_package()
dist, rank, world = P.dist.init_process_group("gloo")
N, T = {N}, {T}
cfg = P.dist.shard_config(P.make_config, rank, world, N, num_steps=T, num_minibatches=2, update_epochs=2, seed=5, total_timesteps=4 * N * T, anneal_lr=False,
                           kernel_flags={kflags})
ctx = P.Context(cfg)
P.dist.bootstrap_comm(ctx, dist, rank, world, P.comm_unique_id, transport="exchange")
# the transport by itself: rank r contributes (r + 1) * pattern; every rank must read back pattern * n (n + 1) / 2, exactly (small integers)
pattern = (np.arange(ctx.P) % 97).astype(np.float32)
for rep in range(5):   # several calls: both slots, flags reused
    ctx.write("GRADS", pattern * (rank + 1) * (rep + 1))
    P.binding._check(P.binding.lib().ppo_allreduce_grads(ctx.h), ctx.h)
    got = ctx.read("GRADS")
    assert np.array_equal(got, pattern * (rep + 1) * (world * (world + 1) // 2)), (rank, rep)
ctx.init_orthogonal(9)
ctx.env_reset()
ctx.train_iteration()
st1 = ctx.stats()     # after the FIRST iteration the job's rollout is, column for column, the single context's: its statistics must be too
ctx.train_iteration()
p = ctx.get_params()
st = ctx.stats()
keys = ("ep_len_mean", "ep_rew_mean", "ep_count", "explained_variance", "global_step", "loss", "pg_loss", "v_loss")
out = dict(rank=rank, timeouts=ctx.comm_exchange_timeouts(), params=p.view(np.uint32).tolist(), loss=st["loss"], steps=st["optimizer_steps"],
           st1=dict((k, st1[k]) for k in keys), st2=dict((k, st[k]) for k in keys))
json.dump(out, open(os.path.join({out!r}, "rank%d.json" % rank), "w"))
dist.barrier()
ctx.close()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 4])
def test_exchange_ranks_equal_single_context(tmp_path, world):
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    P = load_package()
    N, T = 64, 32
    port = _free_port()
    script = tmp_path / "worker.py"
    # More than two ranks on ONE GPU can deadlock for a reason no deployment has (one rank per GPU): the ranks that reach the exchange first
    # spin in 145 workgroups each, 3 x 145 > 256 CUs puts a waiting wave on every CU, and the matrix-core update kernel of the rank they are
    # waiting for needs a CU's WHOLE register file per workgroup -- it can never start, and the waits run out (seen: 30 s, PPO_ERR_COMM).
    # The vector update kernel (ppo_config.kernel_flags: PPO_KERNEL_UPDATE_VECTOR) shares a CU with the waiting waves; the two-rank case keeps
    # the matrix-core kernel under the exchange.
    script.write_text(WORKER.format(root=ROOT, N=N, T=T, out=str(tmp_path), kflags="P.KERNEL_UPDATE_VECTOR" if world > 2 else "0"))
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("exchange worker timed out")
        outs.append(o.decode())
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    res = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(world)]
    assert all(r["timeouts"] == 0 for r in res)
    params = [np.array(r["params"], np.uint32) for r in res]
    for r in range(1, world):
        assert np.array_equal(params[0], params[r]), r          # replicas stay bit-identical
    assert res[0]["steps"] == 2 * 2 * 2
    # one context over all the envs: same init, same envs (global indices), global advantage statistics -> same update up to float noise;
    # permutations are per shard (distributionally equivalent), so only the first iteration's ROLLOUT is comparable sample by sample: compare
    # after ONE optimizer step driven with the whole batch as the minibatch instead
    whole = P.Context(P.make_config(num_envs=N, num_steps=T, num_minibatches=2, update_epochs=2, seed=5, total_timesteps=4 * N * T, anneal_lr=False))
    whole.init_orthogonal(9)
    whole.env_reset()
    whole.train_iteration()
    w1 = whole.stats()
    whole.train_iteration()
    pw = whole.get_params()
    whole.close()
    # job-global statistics (ppo_hip.h, ppo_read_stats): identical on every rank after every update, and after the first iteration -- whose
    # rollout is the single context's, column for column -- equal to what ONE context over all the envs reports (PPO_Discrete.cpp:474-480, 647-648)
    for r in range(1, world):
        assert res[r]["st1"] == res[0]["st1"] and res[r]["st2"] == res[0]["st2"], r
    assert res[0]["st1"]["ep_count"] == w1["ep_count"] > 0
    assert res[0]["st1"]["ep_len_mean"] == w1["ep_len_mean"] and res[0]["st1"]["ep_rew_mean"] == w1["ep_rew_mean"]
    assert res[0]["st1"]["global_step"] == w1["global_step"] == N * T
    assert abs(res[0]["st1"]["explained_variance"] - w1["explained_variance"]) <= 2e-6
    # two full iterations with different minibatch partitions: the trajectories agree in distribution, not element-wise -- bound the distance loosely
    assert np.isfinite(params[0].view(np.float32)).all()
    assert np.abs(params[0].view(np.float32) - pw).max() < 0.05


DEAD_WORKER = r"""
import json, os, sys, time
sys.path.insert(0, {root!r})
import numpy as np
from __graft_entry__ import load_package
P = load_package()
dist, rank, world = P.dist.init_process_group("gloo")
cfg = P.dist.shard_config(P.make_config, rank, world, 64, num_steps=32, num_minibatches=2, update_epochs=2, seed=5, total_timesteps=4 * 64 * 32)
ctx = P.Context(cfg)
P.dist.bootstrap_comm(ctx, dist, rank, world, P.comm_unique_id, transport="exchange")
if rank == 0:   # rank 1 never takes part: rank 0's first call waits out its limit (set to 2 s here; default 30), every later call returns at once,
    # and the failure is REPORTED: the synchronising call after it returns PPO_ERR_COMM
    ctx.comm_set_wait_limit(2.0)
    times, errors = [], []
    for rep in range(3):
        ctx.write("GRADS", np.ones(ctx.P, np.float32))
        t0 = time.perf_counter()
        P.binding._check(P.binding.lib().ppo_allreduce_grads(ctx.h), ctx.h)
        try:
            ctx.sync()
            errors.append("")
        except P.binding.PPOError as ex:
            errors.append(str(ex))
        times.append(time.perf_counter() - t0)
    try:
        ctx.stats()
        stats_error = ""
    except P.binding.PPOError as ex:
        stats_error = str(ex)
    json.dump(dict(times=times, errors=errors, stats_error=stats_error, timeouts=ctx.comm_exchange_timeouts()), open(os.path.join({out!r}, "dead.json"), "w"))
dist.barrier()
ctx.close()
"""


def test_exchange_gives_up_once_when_a_peer_never_arrives(tmp_path):
    """A peer that never calls: the bounded wait ends the first kernel after its limit and marks the communicator dead; the calls after it do not
    wait again, and ppo_sync / ppo_read_stats return PPO_ERR_COMM from then on -- a run whose replicas diverged cannot report a number."""
    port = _free_port()
    script = tmp_path / "dead.py"
    script.write_text(DEAD_WORKER.format(root=ROOT, out=str(tmp_path)))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("worker timed out")
        outs.append(o.decode())
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    res = json.load(open(tmp_path / "dead.json"))
    assert res["timeouts"] != 0
    assert all("gave up waiting for a peer" in e for e in res["errors"]), res
    assert "gave up waiting for a peer" in res["stats_error"], res
    assert 1.0 < res["times"][0] < 6.0, res
    assert res["times"][1] < 0.5 and res["times"][2] < 0.5, res


GENERIC_WORKER = r"""
import json, os, sys
sys.path.insert(0, {root!r})
import numpy as np
from __graft_entry__ import load_package
P = load_package()
dist, rank, world = P.dist.init_process_group("gloo")
cfg = P.dist.shard_config(P.make_config, rank, world, 128, env_kind=P.ENV_SYNTHETIC, dist_kind=P.DIST_MASKED, obs_size=64, head_dims=(3, 2), hidden=256,
                          n_hidden=2, num_steps=16, num_minibatches=2, update_epochs=2, max_episode_steps=10, seed=7, total_timesteps=4 * 128 * 16,
                          compute_dtype={dtype})
ctx = P.Context(cfg)
P.dist.bootstrap_comm(ctx, dist, rank, world, P.comm_unique_id, transport="exchange")
# the transport on this network's gradient (~170 k floats: more 256-element blocks than the kernel's grid, so every workgroup strides)
pattern = (np.arange(ctx.P) % 251).astype(np.float32)
for rep in range(3):
    ctx.write("GRADS", pattern * (rank + 1) * (rep + 1))
    P.binding._check(P.binding.lib().ppo_allreduce_grads(ctx.h), ctx.h)
    assert np.array_equal(ctx.read("GRADS"), pattern * (rep + 1) * (world * (world + 1) // 2)), (rank, rep)
ctx.init_orthogonal(3)
ctx.env_reset()
for _ in range(2):
    ctx.train_iteration()
st = ctx.stats()
json.dump(dict(rank=rank, P=int(ctx.P), timeouts=ctx.comm_exchange_timeouts(), params=ctx.get_params().view(np.uint32).tolist(), loss=st["loss"]),
          open(os.path.join({out!r}, "rank%d.json" % rank), "w"))
dist.barrier()
ctx.close()
"""


@pytest.mark.parametrize("dtype", [0, 1])
def test_exchange_with_a_generic_network(tmp_path, dtype):
    """The exchange under the generic path (any widths; f32 and bf16 storage, the latter with its two streams): a gradient far larger than the
    reference's 9 155 floats, so the exchange kernel's bounded grid walks it in strides; two ranks end bit-identical with no wait run out."""
    port = _free_port()
    script = tmp_path / "generic.py"
    script.write_text(GENERIC_WORKER.format(root=ROOT, out=str(tmp_path), dtype=dtype))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("worker timed out")
        outs.append(o.decode())
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    res = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    assert res[0]["P"] > 256 * 256 and all(r["timeouts"] == 0 for r in res)
    assert res[0]["params"] == res[1]["params"]
    assert np.isfinite(np.array(res[0]["params"], np.uint32).view(np.float32)).all() and np.isfinite(res[0]["loss"])


def test_plain_bench_command_rehearses_two_ranks_on_this_gpu():
    """The driver's plain command, `python bench.py --gpus 2 ...`, with no torchrun environment: bench.py starts the two rank processes itself
    (before anything touches HIP), they rendezvous, bring the transport up, run, and rank 0's ONE line comes back with the transport that ran.
    On this one-GPU box the ranks share the device (--same-device), which only the direct exchange supports -- the line says so."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-device", "--steps", "2", "--warmup", "1", "--envs", "256"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["comm_ranks"] == 2 and d["steps"] == 2 and d["warmup"] == 1
    assert d["transport"] == "exchange" and d["transport_requested"] == "auto" and "same-device" in d["transport_fallback_reason"]
    assert d["config"]["global_batch"] == 2 * 256 * 128 and d["value"] > 0
    assert "dp2" in d["config"]["parallelism"]


def test_bench_single_gpu_line_carries_the_rebased_roofline():
    """N = 1 through the same entry: roofline.frac is the ALGORITHMIC work (53 376 FLOP per sample) over the launch duration against the dense
    f16 matrix peak -- the pipe the kernel issues on; what that pipe executes (the three-product emulation of fp32) sits beside it as frac_executed and
    cannot pass 1; the GAE bar is one object, the transport fields are present and empty, the learning-curve summary rides in train_stats."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "2", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    ro = d["roofline"]
    assert ro["bound"] == "mfma" and ro["peak"] == 2500.0 and 0.03 < ro["frac"] < 0.5 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-9
    assert abs(ro["achieved"] - ro["flops_per_launch"] / (ro["avg_launch_ms"] * 1e-3) / 1e12) < 1e-6 * ro["achieved"]
    assert ro["frac"] < ro["frac_executed"] < 1.0 and 2.5 < ro["executed_flops_per_launch"] / ro["flops_per_launch"] < 3.6 and "limiter" in ro
    assert d["dtype"].startswith("f32 (")
    bar = d["gae_roofline"]["bar"]
    assert bar["target_frac"] == 0.40 and bar["frac_at_config1_back_to_back"] > 0.1 and bar["size_met_from_envs_back_to_back"] in (4096, 8192, 32768, None)
    # what is measured outside the run (counters, the tracer, the curve comparison) is quoted only from a committed profile set whose source fingerprint is
    # this tree's (bench.py: profile_tie); with the sources changed since, those fields are null and the line says why
    tie = d["profiles"]
    if tie["tied"]:
        assert len(d["train_stats"]["learning_curve_parity"]["scenarios"]) >= 2 and ro["traffic"] > 0
        assert bar["frac_at_config1"] > 0.1 and bar["size_met_from_envs"] in (4096, 8192, 32768, 131072, None) and bar["in_trace"]["source"].startswith("profiles/" + tie["tag"])
    else:
        assert "reason" in tie and d["train_stats"]["learning_curve_parity"] is None and ro["traffic"] is None and bar["frac_at_config1"] is None
    assert d["transport"] == "none" and d["comm_ranks"] == 1 and d["transport_ab"] is None
