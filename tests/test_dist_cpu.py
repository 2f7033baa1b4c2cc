"""World-size-2 tests of the data-parallel host logic on CPU (gloo).

What runs here without a GPU: the sharding arithmetic of ppo-libtorch_amd/dist.py, the torch.distributed plumbing bench.py
uses (rendezvous at 127.0.0.1, byte broadcast of the communicator id, max-over-ranks), and the EQUIVALENCE CONTRACT of
SURVEY.md 8(e): two shards that exchange (a) the advantage sums of the minibatch and (b) their 1/M_global-scaled gradients
reproduce the single-process step on the concatenated minibatch.  The arithmetic of a shard is played by the CPU oracle
(checker), the exchange by gloo all-reduces -- the same protocol libppo_hip.so runs over RCCL.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def test_shard_arithmetic():
    P = load_package()
    assert P.dist.shard_envs(32768, 3, 8) == (4096, 12288)
    assert P.dist.shard_envs(8, 0, 1) == (8, 0)
    with pytest.raises(ValueError):
        P.dist.shard_envs(10, 0, 4)
    with pytest.raises(ValueError):
        P.dist.shard_envs(8, 2, 2)
    cfg = P.dist.shard_config(P.make_config, 1, 2, 64, num_steps=16)
    assert (cfg.num_envs, cfg.env_offset, cfg.global_num_envs) == (32, 32, 64)
    # global time-major row (t, e) -> local row of the owning rank
    rows = [0, 5, 63, 64 + 40, 2 * 64 + 31]
    assert P.dist.local_rows_of_global_rows(rows, 16, 64, 0, 2) == [0, 5, 2 * 32 + 31]
    assert P.dist.local_rows_of_global_rows(rows, 16, 64, 1, 2) == [31, 32 + 8]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch
    import oracle as O
    from __graft_entry__ import load_package
    P = load_package()
    dist, r, w = P.dist.init_process_group("gloo")
    assert (r, w) == (rank, world)
    # plumbing used by bench.py
    ident = P.dist.broadcast_bytes(dist, bytes(range(128)) if rank == 0 else None)
    assert ident == bytes(range(128))
    assert P.dist.max_over_ranks(dist, 1.0 + rank) == float(world)
    # plumbing of the one-shot exchange's bootstrap: every rank's handle, in rank order, everywhere; and the post-warm-up agreement check
    handles = P.dist.gather_bytes(dist, bytes([rank]) * 64)
    assert handles == [bytes([r]) * 64 for r in range(world)]
    assert P.dist.all_ranks_agree(dist, True) and not P.dist.all_ranks_agree(dist, rank != world - 1)

    g = O.read_pgld(os.path.join(G, "discrete_t128_n64_seed1.pgld"))
    T, N, U = 128, 64, "u1/"
    n, off = P.dist.shard_envs(N, rank, world)
    net = O.Net.make(4, [2])
    h = g["hparams"]
    hp = O.HParams(gamma=h[1], gae_lambda=h[2], clip_coef=h[3], ent_coef=h[4], vf_coef=h[5], max_grad_norm=h[6], norm_adv=1, clip_vloss=1)
    sl = slice(off, off + n)
    shard = {k: np.ascontiguousarray(g[U + k][:, sl]) for k in ("obs", "actions", "logprobs", "values", "gae_advantages", "gae_returns")}
    params = g[U + "params_before"].copy()
    m, v = np.zeros_like(params), np.zeros_like(params)
    rng = np.random.default_rng(7)
    M_global = 2048
    for step in range(3):
        # a global minibatch with the same number of rows in every shard (equal shards: 8(e))
        t_rows = rng.choice(T, M_global // N, replace=False)
        global_rows = np.array([t * N + e for t in t_rows for e in range(N)])
        local_rows = np.array(P.dist.local_rows_of_global_rows(global_rows, T, N, rank, world))
        assert local_rows.size == M_global // world
        b = lambda k, width=None: shard[k].reshape(T * n, -1) if width else shard[k].reshape(T * n)  # noqa: E731
        # (a) advantage sums of the minibatch: local -> all-reduce
        _, _, local_sums = O.minibatch_grads_shard(net, hp, params, b("obs", 4), b("actions"), b("logprobs"), b("gae_advantages"),
                                                   b("gae_returns"), b("values"), local_rows, M_global)
        sums = torch.tensor(local_sums, dtype=torch.float64)
        dist.all_reduce(sums)
        # (b) shard gradient scaled by 1/M_global (+ its share of the loss means): local -> all-reduce
        grads, stats, _ = O.minibatch_grads_shard(net, hp, params, b("obs", 4), b("actions"), b("logprobs"), b("gae_advantages"),
                                                  b("gae_returns"), b("values"), local_rows, M_global, adv_sums=sums.numpy())
        buf = torch.from_numpy(np.concatenate([grads, np.array([stats[k] for k in O.STAT_NAMES], np.float32)]))
        dist.all_reduce(buf)
        grads = buf[:-6].numpy().copy()
        clipped, total = O.clip_grad_norm(net, grads, hp.max_grad_norm)
        params, m, v = O.adamw_step(params, clipped, m, v, 1e-3, step + 1)
        if rank == 0:
            np.savez(os.path.join(out_dir, "dp_step%d.npz" % step), rows=global_rows, grads=grads, stats=buf[-6:].numpy(), params=params, total=total)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_step_equals_single_process_step(tmp_path, world):
    """2 ranks, and the 8 of BASELINE configs[2] / configs[4] (64 envs -> 8 per rank): the sum over shards of the 1 / M_global-scaled shard
    gradients, formed with the GLOBAL advantage statistics, is the single-process gradient of the concatenated minibatch."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    import oracle as O
    g = O.read_pgld(os.path.join(G, "discrete_t128_n64_seed1.pgld"))
    T, N, U = 128, 64, "u1/"
    net = O.Net.make(4, [2])
    h = g["hparams"]
    hp = O.HParams(gamma=h[1], gae_lambda=h[2], clip_coef=h[3], ent_coef=h[4], vf_coef=h[5], max_grad_norm=h[6], norm_adv=1, clip_vloss=1)
    params = g[U + "params_before"].copy()
    m, v = np.zeros_like(params), np.zeros_like(params)
    for step in range(3):
        dp = np.load(os.path.join(str(tmp_path), "dp_step%d.npz" % step))
        grads, stats = O.minibatch_grads(net, hp, params, g[U + "obs"].reshape(T * N, 4), g[U + "actions"].reshape(T * N), g[U + "logprobs"].ravel(),
                                         g[U + "gae_advantages"].ravel(), g[U + "gae_returns"].ravel(), g[U + "values"].ravel(), dp["rows"])
        # the sum over shards of 1/M-scaled shard gradients IS the gradient of the global minibatch
        assert np.abs(dp["grads"] - grads).max() <= 1e-6 * max(1.0, np.abs(grads).max())
        for i, k in enumerate(O.STAT_NAMES):
            assert abs(dp["stats"][i] - stats[k]) <= 2e-6 * max(1.0, abs(stats[k])), (step, k)
        clipped, total = O.clip_grad_norm(net, grads, hp.max_grad_norm)
        assert abs(total - float(dp["total"])) <= 1e-6 * max(1.0, total)
        params, m, v = O.adamw_step(params, clipped, m, v, 1e-3, step + 1)
        assert np.abs(params - dp["params"]).max() <= 1e-6


class _FakeCtx:
    """Stands where binding.Context stands in dist.start_exchange_checked: fails at the stage `fail_at` names (on the rank that gets one)."""
    def __init__(self, rank, fail_at, log):
        self.rank, self.fail_at, self.log, self.closed = rank, fail_at, log, False
        if fail_at == "create":
            raise RuntimeError("no device")

    def comm_exchange_handle(self):
        if self.fail_at == "handle":
            raise RuntimeError("cannot export")
        return b"handle-of-rank-%d" % self.rank

    def comm_init_exchange(self, handles, rank, world):
        assert handles == [b"handle-of-rank-%d" % r for r in range(world)]
        if self.fail_at == "map":
            raise RuntimeError("hipIpcOpenMemHandle: invalid argument")

    def comm_exchange_timeouts(self):
        return 3 if self.fail_at == "timeout" else 0

    def get_params(self):
        import numpy as np
        return np.full(16, 1.0 + (self.rank if self.fail_at == "diverge" else 0), np.float32)

    def close(self):
        self.closed = True


def _exchange_worker(rank, world, port, out_dir, scenario, bad_rank):
    import json
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    P = load_package()
    dist, rank, world = P.dist.init_process_group("gloo")
    made, lines = [], []
    fail_at = scenario if rank == bad_rank else None

    def make():
        c = _FakeCtx(rank, fail_at, lines)
        made.append(c)
        return c

    def warm(c):
        if c.fail_at == "warm":
            raise RuntimeError("kernel fault")

    got = P.dist.start_exchange_checked(make, warm, dist, rank, world, log=lines.append)
    json.dump(dict(ok=got is not None, closed=[c.closed for c in made], lines=lines), open(os.path.join(out_dir, "x%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scenario", ["fine", "create", "handle", "map", "warm", "timeout", "diverge"])
def test_exchange_start_up_agrees_on_every_rank(tmp_path, scenario):
    """bench.py --transport auto brings the direct exchange up in checked stages (dist.start_exchange_checked).  Whatever stage fails, and on
    whichever single rank, every rank must come out with the same verdict -- through the same sequence of host collectives, so nobody hangs --
    and a rank that gives up closes its context.  Fake contexts over gloo: the failures themselves cannot be provoked on a GPU box."""
    import json
    import torch.multiprocessing as mp
    world, bad_rank = 2, 1
    mp.spawn(_exchange_worker, args=(world, _free_port(), str(tmp_path), scenario, bad_rank), nprocs=world, join=True)
    res = [json.load(open(tmp_path / ("x%d.json" % r))) for r in range(world)]
    want_ok = scenario == "fine"
    assert [r["ok"] for r in res] == [want_ok] * world
    for r, x in enumerate(res):
        if want_ok:
            assert x["closed"] == [False] and x["lines"] == []
        else:
            assert all(x["closed"])                                   # whatever was created has been closed
            if r == bad_rank or scenario in ("timeout", "diverge"):
                assert any("exchange transport failed" in line for line in x["lines"]), x
