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


_STATE = {"backend": "none", "device_group": None, "note": ""}


def init(backend: str, rank: int, world: int):
    """Joins the job.  The data path needs no collective (games are independent), so the only traffic is the timing barrier and a few
    scalar reductions.  The default group is always gloo (host scalars); `backend`:
      "gloo"  barrier over gloo too;
      "nccl"  an RCCL group for the barrier (device-side rendezvous over xGMI); a failure to bring it up is fatal: the rank exits
              non-zero with RCCL's message;
      "auto"  RCCL if it comes up on EVERY rank (agreed over gloo), else gloo with a note - a scaling run is never lost to a
              collective library the data path does not use.
    Returns the backend in use."""
    import torch.distributed as dist
    if world <= 1:
        _STATE.update(backend="none", device_group=None, note="")
        return "none"
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    _STATE.update(backend="gloo", device_group=None, note="")
    if backend in ("nccl", "auto"):
        import torch
        ok, err, grp = 1.0, "", None
        try:
            grp = dist.new_group(backend="nccl")
            t = torch.ones(1, device="cuda")
            dist.all_reduce(t, group=grp)
            torch.cuda.synchronize()
            if int(t.item()) != world:
                raise RuntimeError(f"RCCL all_reduce returned {t.item()} for {world} ranks")
        except Exception as e:   # noqa: BLE001 - whatever RCCL / torch raise
            ok, err = 0.0, repr(e)
        flag = torch.tensor([ok], dtype=torch.float64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)                 # over gloo: every rank learns whether RCCL is up everywhere
        if flag.item() >= 1.0:
            _STATE.update(backend="nccl", device_group=grp)
        elif backend == "nccl":
            raise SystemExit(f"rank {rank}: RCCL did not come up ({err or 'another rank failed'}); --backend gloo runs the same job without it")
        else:
            _STATE.update(note=f"RCCL unavailable ({err or 'on another rank'}): barrier over gloo")
    return _STATE["backend"]


def backend_in_use():
    return _STATE["backend"], _STATE["note"]


def barrier(world: int):
    if world > 1:
        import torch.distributed as dist
        if _STATE["device_group"] is not None:
            dist.barrier(group=_STATE["device_group"])
        dist.barrier()


def max_over_ranks(value: float, world: int, device: str = "cpu") -> float:
    """Timing contract of bench.py: the job takes as long as its slowest rank."""
    if world <= 1:
        return value
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64)          # host scalar over the default (gloo) group
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, world: int, device: str = "cpu") -> float:
    if world <= 1:
        return value
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64)          # host scalar over the default (gloo) group
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(value: float, world: int, device: str = "cpu"):
    """Every rank's value, in rank order (per-GPU rates beside the max-time aggregate)."""
    if world <= 1:
        return [value]
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64)          # host scalar over the default (gloo) group
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]
