"""Environment sharding across the GPUs of one node + the end-of-rollout statistics all-reduce.

Every robosuite environment is an independent world (one MjSim per env object, reference environments/base.py:269), so the path
shards embarrassingly: rank r owns the contiguous GLOBAL env-index block [r*B/G, (r+1)*B/G); per-env RNG
streams are keyed by the global index, so results do not depend on G.  The only collective on the path is one all-reduce (sum) of a
handful of rollout scalars -- RCCL over xGMI on GPUs (backend "nccl"), gloo in the CPU tests.  There is no exchange step inside the
physics, and none is invented here.
"""
from __future__ import annotations

import os

import numpy as np

STAT_FIELDS = ("env_steps", "episodes", "reward_sum", "successes", "diverged")


def env_block(n_total: int, rank: int, world: int):
    """Contiguous block of global env ids owned by `rank` (sizes differ by at most one)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return np.arange(lo, lo + base + (1 if rank < rem else 0), dtype=np.int64)


def dist_env():
    """(rank, local_rank, world) from the torchrun environment; (0, 0, 1) when launched plainly."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_process_group(backend: str | None = None):
    """Join the job's process group (no-op for world size 1).  backend defaults to nccl (= RCCL) with a GPU, gloo without."""
    import torch
    import torch.distributed as dist

    rank, local_rank, world = dist_env()
    if world == 1 or dist.is_initialized():
        return rank, local_rank, world
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kwargs = {}
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        kwargs["device_id"] = torch.device("cuda", local_rank)
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world


class CommUnavailable(RuntimeError):
    """The C-ABI communicator could not be formed on some rank; raised on EVERY rank (the ranks agree before anyone joins)."""


def hip_comm(rank: int, world: int, device: int, unique_id=None, create=None):
    """The C-ABI's own RCCL communicator (include/rsim.h rsim_comm_*, backend.HipComm) for a job that was launched under torch.distributed: rank 0 draws the
    unique id, the job's process group carries its 128 bytes to the other ranks, every rank joins.  Returns None for world size 1.

    Collective-safe: every rank runs the SAME sequence of collectives whatever fails where (round-5 advisor finding: a rank that raised before the broadcast
    left the others blocked in it).  Rank 0 always broadcasts [status byte | 128-byte id]; a failed draw travels as status 1 and every rank raises
    CommUnavailable without calling rsim_comm_create.  After the creation the ranks agree (MAX over ranks of a failure flag) and all of them either keep
    their communicator or free it and raise.  A rank that dies INSIDE ncclCommInitRank can still stall the others in it -- that is RCCL's own rendezvous.
    `unique_id` / `create` replace HipComm.unique_id / HipComm (the gloo tests inject failures through them)."""
    import torch
    import torch.distributed as dist

    from .backend import HipComm

    if world == 1:
        return None
    unique_id = unique_id or HipComm.unique_id
    create = create or HipComm
    dev = torch.device("cuda", device) if dist.get_backend() == "nccl" else torch.device("cpu")
    buf = torch.zeros(1 + HipComm.ID_BYTES, dtype=torch.uint8, device=dev)
    note = ""
    if rank == 0:
        try:
            uid = bytes(unique_id())
            if len(uid) != HipComm.ID_BYTES:
                raise ValueError(f"unique id of {len(uid)} bytes")
            buf[1:].copy_(torch.frombuffer(bytearray(uid), dtype=torch.uint8))
        except Exception as e:   # noqa: BLE001 -- reported to every rank through the status byte
            buf[0] = 1
            note = f"{type(e).__name__}: {e}"
    dist.broadcast(buf, src=0)
    host = buf.cpu().numpy()
    if int(host[0]) != 0:
        raise CommUnavailable("rank 0 could not draw the RCCL unique id" + (f" ({note})" if note else ""))
    comm, err = None, ""
    try:
        comm = create(host[1:].tobytes(), rank, world, device)
    except Exception as e:   # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    if max_over_ranks(0.0 if comm is not None else 1.0, dev) != 0.0:
        if comm is not None and hasattr(comm, "close"):
            comm.close()
        raise CommUnavailable("rsim_comm_create failed on " + (f"this rank ({err})" if err else "another rank"))
    return comm


class RolloutStats:
    """Per-rank rollout accumulators; `allreduce()` returns the whole-job totals (float64 on the wire).  With `comm` (a backend.HipComm, see hip_comm above)
    the reduction is the C-ABI's rsim_allreduce_stats -- the entry a binder without torch.distributed calls -- otherwise the job's torch process group
    (RCCL under backend "nccl", gloo in the CPU tests).  `path` says which one ran."""

    def __init__(self, device="cpu", comm=None):
        import torch

        self.t = torch.zeros(len(STAT_FIELDS), dtype=torch.float64, device=device)
        self.comm = comm
        self.path = "local (world size 1)"

    def add(self, **kw):
        for k, v in kw.items():
            self.t[STAT_FIELDS.index(k)] += float(v)

    def allreduce(self):
        import torch.distributed as dist

        if self.comm is not None:
            self.path = "rsim_allreduce_stats (C-ABI, RCCL)"
            return dict(zip(STAT_FIELDS, self.comm.allreduce(self.t.cpu().numpy(), "sum").tolist()))
        t = self.t.clone()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            self.path = f"torch.distributed ({dist.get_backend()})"
        return dict(zip(STAT_FIELDS, t.tolist()))


def max_over_ranks(x: float, device="cpu") -> float:
    """MAX over ranks of a host scalar (bench timing contract)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
