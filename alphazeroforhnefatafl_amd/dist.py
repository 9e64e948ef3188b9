"""One-process-per-GPU sharding helpers (no collective on the data path: games are independent).

Used by bench.py; covered on CPU with the gloo backend by tests/test_multiproc_cpu.py."""
from __future__ import annotations

import os


def env_rank_world():
    """RANK / LOCAL_RANK / WORLD_SIZE as set by torch.distributed.run (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_base(rank: int, games_per_rank: int) -> int:
    """Global id of the first game of this rank: contiguous ranges, so results equal the 1-GPU run of the same ids."""
    return rank * games_per_rank


def init(backend: str, rank: int, world: int):
    import torch.distributed as dist
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend, rank=rank, world_size=world)


def barrier(world: int):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(value: float, world: int, device: str = "cpu") -> float:
    """Timing contract of bench.py: the job takes as long as its slowest rank."""
    if world <= 1:
        return value
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, world: int, device: str = "cpu") -> float:
    if world <= 1:
        return value
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(value: float, world: int, device: str = "cpu"):
    """Every rank's value, in rank order (per-GPU rates beside the max-time aggregate)."""
    if world <= 1:
        return [value]
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]
