"""Worker for tests/test_multiproc_cpu.py: one rank of a world_size-2 gloo job (CPU).  Each rank runs the ORACLE's MCTS on
its shard of game ids (the GPU path cannot run here) and writes its results; rank 0 also checks the timing reduce."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from alphazeroforhnefatafl_amd import abi, dist as tdist  # noqa: E402
from alphazeroforhnefatafl_amd.abi import TaflMctsParams, TaflState  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    out_dir = sys.argv[1]
    per_rank = int(sys.argv[2])
    rank, local_rank, world = tdist.env_rank_world()
    tdist.init("gloo", rank, world)
    base = tdist.shard_base(rank, per_rank)
    lg = orc.GameLogic(abi.rules.BRANDUBH, 7)
    st = orc.GameState(abi.boards.BRANDUBH, abi.ATTACKER, 64).to_abi()
    states = (TaflState * per_rank)(*[st] * per_rank)
    p = TaflMctsParams(40, 64, 1.0, 9, 0, 0)
    kids, cnt, stats = orc.batch_mcts(lg, states, per_rank, 64, p, base, 64)
    res = {str(base + g): [[kids[g * 64 + j].action, kids[g * 64 + j].visits, kids[g * 64 + j].q.hex()] for j in range(cnt[g])]
           for g in range(per_rank)}
    tdist.barrier(world)
    elapsed = tdist.max_over_ranks(1.0 + rank, world)          # slowest rank = world - 1 -> world
    total = tdist.sum_over_ranks(float(stats.sims), world)
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump({"rank": rank, "world": world, "base": base, "elapsed_max": elapsed, "sims_total": total, "results": res}, f)
    import torch.distributed as dist
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
