#!/usr/bin/env python3
"""bench.py — BASELINE.json's headline metric on MI355X.

One "step" = one pass of the hot path over one batch: `tafl_mcts_run` of S simulations
(select / expand / random-rollout / backup as lock-step HIP kernels) on 65 536 concurrent 11x11 Copenhagen games
per GPU, from the start position (BASELINE.json configs[2]; configs[1] is a parity case, configs[3] is this
workload at N=8).  States are resident in HBM before the timed region; nothing crosses PCIe inside it.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      dominant kernel (k_mcts_rollout): algorithmic HBM bytes per launch / live HIP-event launch time
  cpu_baseline  the oracle (literal C restatement of the reference rules crate + mcts.py arithmetic) timed on the
                host, 1 thread, on a bounded sample of the same workload (+ `all_cores`: one game stream per host core)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GAMES_PER_GPU = 65536
SIDE, WORD_BITS = 11, 128
STATE_BYTES = 64                 # SoA state bytes per 11x11 game (SURVEY.md §8d S_g)
ROLLOUT_BYTES_PER_GAME = 68      # S_g read + 4 B result (register-resident rollout)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(n_sims, c_puct, seed, max_plies, budget_s=20.0):
    """Oracle timed on host cores: same workload (Copenhagen start, S sims/root), bounded sample of games."""
    from alphazeroforhnefatafl_amd import abi
    from alphazeroforhnefatafl_amd.abi import TaflMctsParams
    from oracle import oracle as orc
    lg = orc.GameLogic(abi.rules.COPENHAGEN, SIDE)
    st = orc.GameState(abi.boards.COPENHAGEN, abi.ATTACKER, WORD_BITS).to_abi()
    one = (abi.TaflState * 1)(st)
    p = TaflMctsParams(n_sims, max_plies, c_puct, seed, 0, 0)
    games = sims = plies = 0
    t0 = time.perf_counter()
    while True:
        _, _, stats = orc.batch_mcts(lg, one, 1, WORD_BITS, p, games)
        games += 1
        sims += stats.sims
        plies += stats.rollout_plies
        dt = time.perf_counter() - t0
        if dt > budget_s or games >= 4096:
            break
    # the same oracle on every host core this process may use, one independent game stream per thread;
    # ctypes releases the GIL inside the C call
    import threading
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                    # a container's CPU share (cgroup v2 quota) is what this process can really use
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            ncores = max(1, min(ncores, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    ncores = min(ncores, 32)                # bounded sample: a GPU box hands one GPU's job a 16-CPU share
    mt_budget = budget_s / 2.0
    mt_sims = [0] * ncores

    def worker(t):
        mine = (abi.TaflState * 1)(st)
        g = 1_000_000 + t
        t1 = time.perf_counter()
        while time.perf_counter() - t1 < mt_budget:
            _, _, s2 = orc.batch_mcts(lg, mine, 1, WORD_BITS, p, g)
            mt_sims[t] += s2.sims
            g += ncores

    th = [threading.Thread(target=worker, args=(t,)) for t in range(ncores)]
    t1 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    mt_dt = time.perf_counter() - t1
    return {"value": sims / dt, "unit": "sims/s", "cores": 1, "kind": "port",
            "env_steps_per_sec": plies / dt,
            "all_cores": {"value": sum(mt_sims) / mt_dt, "unit": "sims/s", "cores": ncores, "seconds": round(mt_dt, 1)},
            "sample": f"{games} games x {n_sims} sims of the bench workload (Copenhagen 11x11 start, cap {max_plies}), "
                      f"{dt:.1f} s on 1 host thread (literal C oracle, gcc -O2)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--sims", type=int, default=64, help="MCTS simulations per root (BASELINE config 3: 64/256/1000)")
    ap.add_argument("--games", type=int, default=GAMES_PER_GPU, help="concurrent games per GPU")
    ap.add_argument("--max-plies", type=int, default=512)
    ap.add_argument("--cpuct", type=float, default=1.0)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--board", default="copenhagen11", choices=["copenhagen11", "copenhagen13", "brandubh7"],
                    help="copenhagen11 = the headline workload (BASELINE configs[2]); copenhagen13 = configs[4] (U256 multi-word path)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for the barrier / reductions (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses GPU 0 (needs --backend gloo)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from alphazeroforhnefatafl_amd import dist as tdist
    rank, local_rank, world = tdist.env_rank_world()
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("for --gpus N>1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    tdist.init(args.backend, rank, world)    # RCCL: only for the barrier and the max-reduce of the elapsed time
    red_dev = "cuda" if args.backend == "nccl" else "cpu"

    from alphazeroforhnefatafl_amd import abi
    from alphazeroforhnefatafl_amd.engine import KC_MCTS_BACKUP, KC_MCTS_ROLLOUT, KC_MCTS_TREE, BatchedGameLogic

    stream = torch.cuda.Stream()
    rules_, fen_, side_, wb_ = {"copenhagen11": (abi.rules.COPENHAGEN, abi.boards.COPENHAGEN, 11, 128),
                                "copenhagen13": (abi.rules.COPENHAGEN, abi.boards.COPENHAGEN13, 13, 256),
                                "brandubh7": (abi.rules.BRANDUBH, abi.boards.BRANDUBH, 7, 64)}[args.board]
    logic = BatchedGameLogic(rules_, side_, wb_, device=local_rank, stream=stream.cuda_stream)
    G = args.games
    batch = logic.new_batch(G, fen_)                          # synthetic data: every game at the start position
    batch.mcts_reserve(args.sims)
    base = tdist.shard_base(rank, G)                          # contiguous global game-id shards, no collective on the data path

    def step():
        batch.mcts_run(args.sims, args.cpuct, args.seed, args.max_plies, game_id_base=base)

    def barrier():
        torch.cuda.synchronize()
        tdist.barrier(world)
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    logic.timing_reset()
    logic.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    logic.timing_enable(False)
    stats = batch.mcts_stats()                                # counters of the last step
    roll_ms, roll_n = logic.timing_get(KC_MCTS_ROLLOUT)
    tree_ms, tree_n = logic.timing_get(KC_MCTS_TREE)
    bk_ms, bk_n = logic.timing_get(KC_MCTS_BACKUP)

    elapsed = tdist.max_over_ranks(elapsed, world, device=red_dev)

    total_sims = float(world) * G * args.sims * args.steps
    plies_per_step = float(stats.rollout_plies)
    total_plies = tdist.sum_over_ranks(plies_per_step, world, device=red_dev) * args.steps
    if rank == 0:
        # playouts actually executed by k_mcts_rollout in the last step: consumed ones + mispredicted speculative ones
        executed = int(stats.rollouts - stats.spec_hits + stats.spec_issued)
        roll_s_per_step = roll_ms * 1e-3 / args.steps
        sg_bytes = {7: 48, 11: 64, 13: 96}[side_]
        d_bar = stats.tree_depth_sum / max(stats.sims, 1)
        c_bar = stats.children_scanned / max(stats.tree_depth_sum, 1)
        bytes_per_sim = d_bar * (64 + 16 * c_bar) + 32 + 2 * sg_bytes + 4        # SURVEY.md §8d formula (select + expand + rollout + backup)
        # default path: ONE kernel (k_mcts_fused) runs tree phase and playouts, so its algorithmic bytes are the whole simulation's
        # plus 68 B for every mispredicted speculative playout; TAFL_MCTS_FUSED=0 times k_mcts_rollout alone (68 B per playout)
        fused = tree_n == 0
        kname = "k_mcts_fused" if fused else "k_mcts_rollout"
        if fused:
            alg_bytes_per_step = bytes_per_sim * float(stats.sims) + (sg_bytes + 4) * max(0, executed - int(stats.rollouts))
        else:
            alg_bytes_per_step = (sg_bytes + 4) * executed
        achieved = alg_bytes_per_step / roll_s_per_step / 1e9 if roll_s_per_step > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % kname)
        if os.path.exists(tpath) and G == GAMES_PER_GPU and args.board == "copenhagen11" and args.sims == 64:
            with open(tpath) as f:
                traffic = json.load(f).get("hbm_bytes_per_step")        # rocprofv3 PMC passes of this same command (profiles/)
        out = {
            "metric": "mcts_sims_per_sec", "value": total_sims / elapsed, "unit": "sims/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[2]" if args.board == "copenhagen11" and G == GAMES_PER_GPU else "variant") +
                                   f": full MCTS (select/expand/random-rollout/backup), {G} concurrent {side_}x{side_} "
                                   f"{args.board} games per GPU from the start position",
                       "games_per_gpu": G, "sims_per_root": args.sims, "max_rollout_plies": args.max_plies,
                       "c_puct": args.cpuct, "seed": args.seed, "sharding": f"game-id ranges x{world}, no collectives"},
            "env_steps_per_sec": total_plies / elapsed,
            "plies_per_rollout": plies_per_step / max(stats.rollouts, 1),
            "mcts": {"sims": int(stats.sims), "rollouts": int(stats.rollouts), "terminal_hits": int(stats.terminal_hits),
                     "mean_select_depth": d_bar, "mean_children_scanned": c_bar, "faults": int(stats.faults), "spec_issued": int(stats.spec_issued), "spec_hits": int(stats.spec_hits),
                     "reason_hist": [int(x) for x in stats.reason_hist],
                     "algorithmic_bytes_per_sim": bytes_per_sim,
                     "hbm_frac_sims": (total_sims / elapsed / world) * bytes_per_sim / (HBM_PEAK_GBS * 1e9)},
            "kernels_ms": {kname: {"avg": roll_ms / max(roll_n, 1), "launches": int(roll_n), "total_per_step": roll_ms / args.steps},
                           "k_mcts_tree": {"avg": tree_ms / max(tree_n, 1), "launches": int(tree_n), "total_per_step": tree_ms / args.steps},
                           "k_mcts_tree(final backup)": {"avg": bk_ms / max(bk_n, 1), "launches": int(bk_n)}},
            "roofline": {"kernel": kname, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes": alg_bytes_per_step, "kernel_ms": roll_s_per_step * 1e3, "per": "step (all %s launches)" % kname,
                         "launch_avg_ms": roll_ms / max(roll_n, 1), "playouts_executed": executed,
                         "note": "register-resident playouts (68 algorithmic bytes each) + the tree phase's node / edge records; the binding limit is integer "
                                 "VALU issue (about 80 % of a 1-instruction-per-4-cycles-per-SIMD issue model; instruction-class rates in "
                                 "profiles/r01_valu_rates), see DESIGN.md section 6"},
        }
        if world == 1 and not args.no_cpu_baseline and args.board == "copenhagen11":
            out["cpu_baseline"] = cpu_baseline(args.sims, args.cpuct, args.seed, args.max_plies)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
