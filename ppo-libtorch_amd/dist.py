"""Host logic of the data-parallel path (no reference counterpart).

One process per GPU: rank r owns the contiguous env shard [r * N/W, (r+1) * N/W) and its slice of the time-major rollout
buffers; weights are replicated (same init seed); the only data-path collective is ONE all-reduce of the flat gradient per
optimizer step (plus one tiny all-reduce of the per-minibatch advantage sums per update), done inside libppo_hip.so on its own
RCCL communicator.  torch.distributed is used here for what it is good at as plumbing: rendezvous, broadcasting the RCCL unique
id, barriers and the max-over-ranks of the timing.
"""
import os


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


def max_over_ranks(dist, value):
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
