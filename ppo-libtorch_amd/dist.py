"""Host logic of the data-parallel path (no reference counterpart).

One process per GPU: rank r owns the contiguous env shard [r * N/W, (r+1) * N/W) and its slice of the time-major rollout
buffers; weights are replicated (same init seed); the only data-path collective is ONE all-reduce of the flat gradient per
optimizer step (plus one tiny all-reduce of the per-minibatch advantage sums per update), done inside libppo_hip.so on its own
RCCL communicator.  torch.distributed is used here for what it is good at as plumbing: rendezvous, broadcasting the RCCL unique
id, barriers and the max-over-ranks of the timing.
"""
import os


def enable_dmabuf_ipc():
    """HIP IPC between processes (RCCL's intra-node transport and the direct exchange's peer buffers) needs dmabuf IPC on hosts whose driver supports
    nothing else; without it hipIpcGetMemHandle / ncclCommInitRank fail with "invalid argument".  It must be in the environment before the process's
    first HIP call: init_process_group() sets it for world > 1 (the multi-rank path only -- importing this module changes nothing for a single-GPU
    user), and a launcher should put it into the ranks' environment itself (bench.py does)."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def shard_envs(global_num_envs, rank, world):
    """Contiguous equal shards.  Returns (num_envs, env_offset)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank %d of %d" % (rank, world))
    if global_num_envs % world != 0:
        raise ValueError("global_num_envs=%d is not divisible by world=%d (equal shards are required: the gradient of the global "
                         "minibatch is the plain sum of the shard gradients only then)" % (global_num_envs, world))
    n = global_num_envs // world
    return n, rank * n


def shard_config(make_config, rank, world, global_num_envs, **kw):
    """The ppo_config of rank `rank`: its shard of the envs, the global env count and its global env offset (the offset keeps
    per-env RNG streams and the reference's 'env 0 is reset twice' quirk tied to GLOBAL env indices)."""
    n, off = shard_envs(global_num_envs, rank, world)
    kw = dict(kw)
    kw.update(num_envs=n, env_offset=off, global_num_envs=global_num_envs)
    return make_config(**kw)


def local_rows_of_global_rows(global_rows, num_steps, global_num_envs, rank, world):
    """Maps rows of the global flattened batch [T * N] (time-major) to rows of rank's flattened batch [T * N/W]; returns the local
    rows of the global rows this rank owns (same order)."""
    n, off = shard_envs(global_num_envs, rank, world)
    out = []
    for g in global_rows:
        t, e = divmod(int(g), global_num_envs)
        if off <= e < off + n:
            out.append(t * n + (e - off))
    return out


def init_process_group(backend="gloo"):
    """torch.distributed rendezvous from the torchrun environment (RANK / WORLD_SIZE / MASTER_*); returns (dist, rank, world)."""
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost") and os.path.exists("/sys/class/net/lo"):
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # one node: gloo must not depend on the host name resolving
    if world > 1:
        enable_dmabuf_ipc()
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist, rank, world


def broadcast_bytes(dist, payload, src=0):
    """Broadcasts a bytes object made on rank `src` (used for the RCCL unique id)."""
    box = [payload]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def gather_bytes(dist, payload):
    """Every rank's bytes object, in rank order, on every rank."""
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, payload)
    return out


def bootstrap_comm(ctx, dist, rank, world, make_unique_id, transport="rccl"):
    """Gives `ctx` its communicator.
    transport "rccl":     rank 0 makes the RCCL unique id, everybody receives it, everybody joins (ncclCommInitRank).
    transport "exchange": the one-shot direct exchange (ppo_hip.h): every rank exports the IPC handle of its exchange buffer, the handles are
                          gathered in rank order, every rank maps its peers' buffers.  No RCCL at all."""
    if world == 1:
        return
    if "HSA_ENABLE_IPC_MODE_LEGACY" not in os.environ:
        # a caller that ran its own torch.distributed rendezvous: `ctx` exists, so HIP is already initialised and setting the variable now may come too late
        import warnings
        warnings.warn("HSA_ENABLE_IPC_MODE_LEGACY was not set before the context was created: on hosts that only support dmabuf IPC, hipIpcGetMemHandle / "
                      "ncclCommInitRank fail with 'invalid argument'.  Export HSA_ENABLE_IPC_MODE_LEGACY=0 in the ranks' environment (INTEGRATION.md).")
    enable_dmabuf_ipc()
    if transport == "exchange":
        handles = gather_bytes(dist, ctx.comm_exchange_handle())
        ctx.comm_init_exchange(handles, rank, world)
        return
    if transport != "rccl":
        raise ValueError("transport must be 'rccl' or 'exchange'")
    ident = broadcast_bytes(dist, make_unique_id() if rank == 0 else None, src=0)
    ctx.comm_init(ident, rank, world)


def all_ranks_agree(dist, ok):
    """True iff `ok` is true on every rank."""
    import torch
    t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item() > 0.5)


def values_of_ranks(dist, values):
    """A short list of floats from every rank, in rank order, on every rank (per-rank timings of a bench line)."""
    import torch
    mine = torch.tensor([float(v) for v in values], dtype=torch.float64)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [[float(x) for x in t.tolist()] for t in out]


def max_over_ranks(dist, value):
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def start_exchange_checked(make_ctx, warm, dist, rank, world, log=None):
    """Brings a context up on the one-shot direct exchange in checked stages -- the exchange has never run on a node before the job that uses
    it.  Every rank walks the SAME sequence of host collectives whether or not its own stage failed (a rank that left the sequence early
    would leave the others in a mismatched collective):
        1. create the context and export its exchange handle        -> all-gather of the handles (None from a rank that failed)
        2. map the peers' buffers                                   -> agreement
        3. `warm(ctx)` (initialise, reset, warm-up iterations)      -> all-gather of a digest of the parameters (None: failed or a wait ran out)
        4. every digest present and equal (replicas bit-identical)  -> agreement
    Returns the context, or None when any rank failed any stage (this rank's context is then closed)."""
    import hashlib
    import sys
    log = log or sys.stderr.write
    if world > 1:
        enable_dmabuf_ipc()   # before make_ctx()'s first HIP call (a no-op when the launcher exported it, as bench.py does)
    c, err, handle = None, None, None
    try:
        c = make_ctx()
        handle = c.comm_exchange_handle()
    except Exception as ex:
        err = ex
    handles = gather_bytes(dist, handle)
    if err is None:
        try:
            if any(h is None for h in handles):
                raise RuntimeError("a peer could not export its exchange buffer")
            c.comm_init_exchange(handles, rank, world)
        except Exception as ex:   # e.g. IPC handles cannot be opened on this node
            err = ex
    if all_ranks_agree(dist, err is None):
        digest = None
        try:
            warm(c)
            if c.comm_exchange_timeouts() == 0:
                digest = hashlib.sha256(c.get_params().tobytes()).digest()
        except Exception as ex:
            err = ex
        digests = gather_bytes(dist, digest)
        if digest is None or len(set(digests)) != 1:
            err = err or RuntimeError("replicas differ or a wait ran out after the warm-up")
        if all_ranks_agree(dist, err is None):
            return c
    if err is not None:
        log("rank %d: exchange transport failed (%r)\n" % (rank, err))
    if c is not None:
        c.close()
    return None
