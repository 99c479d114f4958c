"""Multi-GPU: environments are independent (each owns its vessel, path and obstacles,
/root/reference/gym_auv/environment.py:86-89), so the batch shards by contiguous
environment-index blocks, one process per GPU, with NO collective on the step() path.
The only exchange is an all_gather of per-environment episode statistics for reporting
(12-16 B per env, once per reporting interval) over RCCL/xGMI (`backend="nccl"` is RCCL on
ROCm) or gloo on CPU.  World seeds are `base + global_env_index`, so results do not depend
on how many GPUs the batch is spread over."""
import os
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (as set by
    torch.distributed.run).  Returns (rank, world_size, local_rank); no-op for world_size 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # AUV_DIST_BACKEND=gloo rehearses the multi-rank path where RCCL cannot run
            # (several ranks sharing one GPU, or no GPU at all)
            backend = os.environ.get("AUV_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of global environment indices owned by `rank`; blocks differ
    by at most one env and cover [0, n_total) exactly."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_episode_stats(stats: Dict[str, torch.Tensor], n_max: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """all_gather the per-env episode statistics of every rank (rank order == global env
    order).  `stats` maps name -> 1-D tensor of this rank's envs.  Shards may differ in
    length by one; they are padded to `n_max` for the collective and trimmed afterwards."""
    keys = sorted(stats)
    local = torch.stack([stats[k].to(torch.float32) for k in keys], dim=1)   # [n_local, F]
    if not (dist.is_available() and dist.is_initialized()):
        return {k: local[:, i] for i, k in enumerate(keys)}
    world = dist.get_world_size()          # (an initialised group of ONE rank still goes through the collective)
    src_device = local.device
    if dist.get_backend() == "gloo":
        local = local.cpu()              # gloo collectives run on host tensors
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    counts = [int(c.item()) for c in counts]
    n_pad = n_max or max(counts)
    padded = torch.zeros((n_pad, local.shape[1]), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded)
    full = torch.cat([o[:c] for o, c in zip(out, counts)], dim=0).to(src_device)
    return {k: full[:, i] for i, k in enumerate(keys)}


def max_over_ranks(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(values, device) -> list:
    """Every rank's list of floats, in rank order ([[...] per rank]); reporting only."""
    if not (dist.is_available() and dist.is_initialized()):
        return [list(map(float, values))]
    t = torch.tensor(list(map(float, values)), dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(x) for x in o.cpu()] for o in out]


def backend_name() -> str:
    """The collective backend in use ("nccl" is RCCL on ROCm), or "none" for a single process."""
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else "none"


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
