"""Multi-GPU: one process per GPU, queries sharded, ONE collective -- the all-reduce of the
nFiles-long int64 hits vector (RCCL over xGMI when the backend is "nccl"; gloo on CPU for
tests).  Exactness: hits[] is a sum over queries of non-negative integers, so any partition
of the queries gives the identical vector."""
import os


def shard_bounds(n, world, rank):
    """Contiguous slab [lo, hi) of n queries for `rank` (slabs differ by at most one query)."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_from_env(backend=None):
    """torch.distributed process group from RANK / WORLD_SIZE / MASTER_* (torchrun contract)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    # IGD_DIST_FORCE=1: a process group also for ONE rank -- the collective path (librccl next to libigd_hip.so in one
    # process, an int64 SUM all-reduce on the engine's stream) can then be exercised on a single-GPU box
    force = os.environ.get("IGD_DIST_FORCE", "") not in ("", "0")
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend is None:
            # IGD_DIST_BACKEND=gloo: control-flow tests of the N>1 path on a single-GPU box
            backend = os.environ.get("IGD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def allreduce_hits(hits_tensor):
    """In-place SUM all-reduce of the per-rank hits vector (int64).  No-op for world size 1."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("IGD_DIST_FORCE", "") not in ("", "0")):
        dist.all_reduce(hits_tensor, op=dist.ReduceOp.SUM)
    return hits_tensor


def ranks_seen():
    """World size as the process group reports it (1 without one): goes into the bench line of an N > 1 run."""
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def gather_strings(text):
    """Every rank's string on every rank (device ordinal / bus id of each rank for the bench line)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [text]
    out = [None] * dist.get_world_size()
    try:
        dist.all_gather_object(out, text)
    except Exception as e:                                  # a label for the bench line must never cost the run
        return [text + " (all_gather_object failed: %s)" % e]
    return out
