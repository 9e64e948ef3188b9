#!/usr/bin/env python3
"""bench.py — BASELINE.json's headline metric on MI355X.

One "step" = one pass of the hot path over one batch: `tafl_mcts_run` of S simulations
(select / expand / random-rollout / backup as lock-step HIP kernels) on 65 536 concurrent 11x11 Copenhagen games
per GPU, from the start position (BASELINE.json configs[2]; configs[3] is this workload at N=8, 524 288 games).
States are resident in HBM before the timed region; nothing crosses PCIe inside it.

  python bench.py --gpus N --steps K --warmup W

N > 1 runs one process per GPU, each with its own shard of global game ids and NO collective on the data path (a barrier
and a max-reduce of the elapsed time only).  Two ways to get there:
  * the driver's:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
  * directly:      python bench.py --gpus N ...   (no RANK/WORLD_SIZE in the environment): this process starts N fresh children
                   with RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set BEFORE anything touches torch or HIP, waits for them and
                   exits with the worst child's code.  `--single-device` (rehearsal on a 1-GPU box: every rank on GPU 0,
                   gloo for the barrier) exercises the same code path.

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline      dominant kernel: algorithmic HBM bytes per step / live HIP-event time of its launches
  cpu_baseline  the oracle (literal C restatement of the reference rules crate + mcts.py arithmetic) timed on the
                host, 1 thread, on a bounded sample of the same workload (+ `all_cores`, + BASELINE configs[0])
  variants      the rest of SURVEY.md section 8d, measured after the headline timed region (N=1 only): S=256 / S=1000,
                13x13 (configs[4]), and the streamed configs[1] kernels at 4 096 and 65 536 games, each with its roofline
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GAMES_PER_GPU = 65536
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 2   # wave64 VALU instructions per second: 1 024 SIMD-32 x 2.4 GHz, 2 cycles per instruction (MI355X_MICROARCH.md "Wave scheduling")
SG_BYTES = {7: 48, 11: 64, 13: 96}   # SoA state bytes per game (SURVEY.md section 8d S_g)
BOARDS = {"copenhagen11": ("COPENHAGEN", "COPENHAGEN", 11, 128), "copenhagen13": ("COPENHAGEN", "COPENHAGEN13", 13, 256),
          "brandubh7": ("BRANDUBH", "BRANDUBH", 7, 64)}


# ----------------------------------------------------------------------------------------------------------------------
# N > 1 without torchrun: fresh child processes, one per GPU
# ----------------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n: int, argv) -> int:
    """Start n children of this script (rank r -> GPU r) and wait.  Nothing in this process has imported torch or called
    HIP, so every child initialises its GPU in a fresh process (never exec from a GPU-initialised process)."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   TAFL_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    deadline = time.time() + 3000
    for p in procs:
        try:
            code = p.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            code = 124
        rc = rc or code
    if rc:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


# ----------------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle is the checker / the baseline only: nothing below is on the product path)
# ----------------------------------------------------------------------------------------------------------------------
def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _oracle_build() -> str:
    flags = "-O2 -std=gnu11 -fPIC -ffp-contract=off -fno-fast-math"
    try:
        for line in open(os.path.join(ROOT, "oracle", "Makefile")):
            if line.startswith("CFLAGS"):
                flags = line.split("=", 1)[1].strip()
        cc = subprocess.run(["gcc", "--version"], capture_output=True, text=True).stdout.splitlines()[0]
    except Exception:
        cc = "gcc"
    return f"{cc}; {flags}"


def _host_cores() -> int:
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                    # a container's CPU share (cgroup v2 quota) is what this process can really use
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            ncores = max(1, min(ncores, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return min(ncores, 32)                  # bounded sample: a GPU box hands one GPU's job a 16-CPU share


def cpu_baseline(n_sims, c_puct, seed, max_plies, budget_s=16.0):
    """Oracle timed on host cores: same workload (Copenhagen start, S sims/root), bounded sample of games."""
    import threading

    from alphazeroforhnefatafl_amd import abi
    from alphazeroforhnefatafl_amd.abi import TaflMctsParams
    from oracle import oracle as orc
    side, wb = 11, 128
    lg = orc.GameLogic(abi.rules.COPENHAGEN, side)
    st = orc.GameState(abi.boards.COPENHAGEN, abi.ATTACKER, wb).to_abi()
    one = (abi.TaflState * 1)(st)
    p = TaflMctsParams(n_sims, max_plies, c_puct, seed, 0, 0)
    games = sims = plies = 0
    t0 = time.perf_counter()
    while True:
        _, _, stats = orc.batch_mcts(lg, one, 1, wb, p, games)
        games += 1
        sims += stats.sims
        plies += stats.rollout_plies
        dt = time.perf_counter() - t0
        if dt > budget_s * 0.5 or games >= 4096:
            break
    # the same oracle on every host core this process may use, one independent game stream per thread;
    # ctypes releases the GIL inside the C call
    ncores = _host_cores()
    mt_budget = budget_s * 0.3
    mt_sims = [0] * ncores

    def worker(t):
        mine = (abi.TaflState * 1)(st)
        g = 1_000_000 + t
        t1 = time.perf_counter()
        while time.perf_counter() - t1 < mt_budget:
            _, _, s2 = orc.batch_mcts(lg, mine, 1, wb, p, g)
            mt_sims[t] += s2.sims
            g += ncores

    th = [threading.Thread(target=worker, args=(t,)) for t in range(ncores)]
    t1 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    mt_dt = time.perf_counter() - t1
    # BASELINE.json configs[0] / BASELINE.md C1: single 7x7 Brandubh game, 1000-sim random-rollout MCTS, 1 thread
    blg = orc.GameLogic(abi.rules.BRANDUBH, 7)
    bst = (abi.TaflState * 1)(orc.GameState(abi.boards.BRANDUBH, abi.ATTACKER, 64).to_abi())
    bp = TaflMctsParams(1000, 256, 1.0, 0, 0, 0)
    t2 = time.perf_counter()
    reps = bs = 0
    while True:
        _, _, s3 = orc.batch_mcts(blg, bst, 1, 64, bp, reps)
        reps += 1
        bs += s3.sims
        if time.perf_counter() - t2 > budget_s * 0.15 or reps >= 64:
            break
    c1_dt = time.perf_counter() - t2
    return {"value": sims / dt, "unit": "sims/s", "cores": 1, "kind": "port",
            "env_steps_per_sec": plies / dt,
            "cpu_model": _cpu_model(), "compiler": _oracle_build(),
            "all_cores": {"value": sum(mt_sims) / mt_dt, "unit": "sims/s", "cores": ncores, "seconds": round(mt_dt, 1)},
            "config1_brandubh_1000sims": {"value": bs / c1_dt, "unit": "sims/s", "cores": 1,
                                          "sample": f"{reps} searches of 1000 sims (cap 256, seed 0) from the Brandubh start, {c1_dt:.1f} s"},
            "sample": f"{games} games x {n_sims} sims of the bench workload (Copenhagen 11x11 start, cap {max_plies}), "
                      f"{dt:.1f} s on 1 host thread (literal C oracle)"}


# ----------------------------------------------------------------------------------------------------------------------
# measurement helpers (GPU)
# ----------------------------------------------------------------------------------------------------------------------
def mcts_bytes_per_sim(stats, side):
    """d*(64+16*c) + 32 + 2*S_g + 4 bytes per simulation (select + expand + rollout + backup)."""
    d_bar = stats.tree_depth_sum / max(stats.sims, 1)
    c_bar = stats.children_scanned / max(stats.tree_depth_sum, 1)
    return d_bar * (64 + 16 * c_bar) + 32 + 2 * SG_BYTES[side] + 4, d_bar, c_bar


def timed_mcts(logic, batch, sims, cpuct, seed, cap, base, steps, warmup, sync, flags=0):
    """W warm-up + K timed mcts_run steps on an existing batch; returns (elapsed_s, stats, kernel-class timings)."""
    from alphazeroforhnefatafl_amd.engine import KC_MCTS_ROLLOUT, KC_MCTS_TREE
    for _ in range(warmup):
        batch.mcts_run(sims, cpuct, seed, cap, game_id_base=base, flags=flags)
    sync()
    logic.timing_reset()
    logic.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        batch.mcts_run(sims, cpuct, seed, cap, game_id_base=base, flags=flags)
    sync()
    elapsed = time.perf_counter() - t0
    logic.timing_enable(False)
    kt = {"rollout": logic.timing_get(KC_MCTS_ROLLOUT), "tree": logic.timing_get(KC_MCTS_TREE),
          "rollout_union": logic.timing_get_union(KC_MCTS_ROLLOUT), "tree_union": logic.timing_get_union(KC_MCTS_TREE)}
    return elapsed, batch.mcts_stats(), kt


def mcts_roofline(stats, kt, steps, side, traffic=None, traffic_source=None):
    """Roofline object of the dominant MCTS kernel of a run (the fused kernel carries whole simulations; in the two-kernel
    pipeline the playout kernel moves S_g + 4 bytes per executed playout)."""
    roll_ms, roll_n = kt["rollout"]
    tree_ms, tree_n = kt["tree"]
    executed = int(stats.rollouts - stats.spec_hits + stats.spec_issued)      # consumed + mispredicted speculative playouts
    sg = SG_BYTES[side]
    bps, d_bar, c_bar = mcts_bytes_per_sim(stats, side)
    fused = tree_n == 0
    kname = "k_mcts_fused" if fused else "k_mcts_rollout"
    alg = bps * float(stats.sims) + (sg + 4) * max(0, executed - int(stats.rollouts)) if fused else float((sg + 4) * executed)
    # the partitions' launches overlap on their streams: the kernel's time is the UNION of the launch intervals (live HIP events on the
    # streams the kernel runs on), launch_avg_ms the plain average of the launch durations, overlap_factor = sum / union
    union_ms, sum_ms = kt.get("rollout_union", (roll_ms, roll_ms))
    s_per_step = union_ms * 1e-3 / steps
    achieved = alg / s_per_step / 1e9 if s_per_step > 0 else 0.0
    r = {"kernel": kname, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "traffic": traffic, "algorithmic_bytes": alg, "kernel_ms": s_per_step * 1e3, "per": "step (union of the %s launch intervals)" % kname,
         "launch_avg_ms": roll_ms / max(roll_n, 1), "launches_per_step": roll_n / steps, "overlap_factor": sum_ms / union_ms if union_ms > 0 else 0.0,
         "playouts_executed": executed}
    if traffic is not None:
        r["traffic_source"] = traffic_source
    return r, kname, bps, d_bar, c_bar


def streamed_variant(logic, fen, side, G):
    """BASELINE configs[1] workload (games advanced by (i mod 64) seeded random plies): kernel-only HIP-event times of the
    streamed entry points with their algorithmic HBM rates (372 B / 136 B / 68 B per game on 11x11)."""
    from alphazeroforhnefatafl_amd.engine import KC_MOVEGEN, KC_ROLLOUT, KC_STEP
    b = logic.new_batch(G, fen)
    plies = (C.c_uint32 * G)(*[i % 64 for i in range(G)])
    b.random_advance(1, plies, 0)
    ranks = (C.c_uint32 * G)(*[(i * 2654435761) & 0x3FFFFFFF for i in range(G)])
    sg, mask_b = SG_BYTES[side], 4 * logic.mask_words
    out = {}

    def rec(name, ms, k, bytes_per_game, unit="games/s"):
        us = ms / max(k, 1) * 1e3
        gbs = G * bytes_per_game / (us * 1e-6) / 1e9 if us > 0 else 0.0
        out[name] = {"us": us, "rate": G / (us * 1e-6) if us > 0 else 0.0, "unit": unit,
                     "roofline": {"bound": "hbm", "algorithmic_bytes_per_game": bytes_per_game, "achieved": gbs, "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}}

    b.iter_plays(want_masks=False); b.iter_plays(want_masks=True)      # warm-up (code load)
    logic.timing_enable(True)
    logic.timing_reset()
    for _ in range(5):
        b.iter_plays(want_masks=False)
    rec("movegen_counts", *logic.timing_get(KC_MOVEGEN), sg + 4)
    logic.timing_reset()
    for _ in range(5):
        b.iter_plays(want_masks=True)
    rec("movegen_masks", *logic.timing_get(KC_MOVEGEN), sg + 4 + mask_b)
    states = b.download()
    scratch = logic.new_batch(G)
    scratch.upload(states)
    scratch.do_kth_play(ranks)                                          # warm-up
    ms_k = k_k = 0
    plays = None
    for _ in range(3):
        scratch.upload(states)
        logic.timing_reset()
        plays, _ = scratch.do_kth_play(ranks)
        m1, k1 = logic.timing_get(KC_STEP)
        ms_k += m1; k_k += k1
    rec("step_kth", ms_k, k_k, 2 * sg + 4 + 4 + 40, "steps/s")
    scratch.close()
    ms_s = k_s = 0
    for _ in range(3):
        b.upload(states)
        logic.timing_reset()
        b.do_play(plays)
        m1, k1 = logic.timing_get(KC_STEP)
        ms_s += m1; k_s += k1
    rec("step", ms_s, k_s, 2 * sg + 4 + 40, "steps/s")
    b.upload(states)
    b.rollout(3, 0, 512, 0)
    logic.timing_reset()
    res = b.rollout(3, 0, 512, 0)
    ms_r, k_r = logic.timing_get(KC_ROLLOUT)
    logic.timing_enable(False)
    total_plies = sum(r.plies for r in res)
    rec("rollout", ms_r, k_r, sg + 4, "playouts/s")
    out["rollout"]["plies_per_sec"] = total_plies / (ms_r / max(k_r, 1) * 1e-3) if ms_r > 0 else 0.0
    b.close()
    return out


def run_variants(args, torch):
    """SURVEY.md section 8d beyond the headline line; every entry measured here, now, on this GPU."""
    from alphazeroforhnefatafl_amd import abi
    from alphazeroforhnefatafl_amd.engine import BatchedGameLogic
    out = {}

    def sync():
        torch.cuda.synchronize()

    def mcts_variant(board, sims, steps, warmup, mixed=False):
        rn, bn, side, wb = BOARDS[board]
        logic = BatchedGameLogic(getattr(abi.rules, rn), side, wb, device=0)
        batch = logic.new_batch(GAMES_PER_GPU, getattr(abi.boards, bn))
        if mixed:       # the configs[1] input recipe: game i advanced by (i mod 64) seeded random plies - what self-play batches look like
            batch.random_advance(1, (C.c_uint32 * GAMES_PER_GPU)(*[i % 64 for i in range(GAMES_PER_GPU)]), 0)
        batch.mcts_reserve(sims)
        el, st, kt = timed_mcts(logic, batch, sims, args.cpuct, args.seed, args.max_plies, 0, steps, warmup, sync)
        roof, kname, bps, d_bar, c_bar = mcts_roofline(st, kt, steps, side)
        assert st.sims == GAMES_PER_GPU * sims and st.faults == 0, (board, sims, st.sims, st.faults)
        r = {"workload": f"{GAMES_PER_GPU} x {side}x{side} {board}, S={sims}, cap {args.max_plies}" + (", positions after (i mod 64) random plies" if mixed else ", start position"),
             "value": GAMES_PER_GPU * sims * steps / el,
             "unit": "sims/s", "ms_per_step": el / steps * 1e3, "env_steps_per_sec": float(st.rollout_plies) * steps / el,
             "mean_select_depth": d_bar, "mean_children_scanned": c_bar, "spec_hit_rate": st.spec_hits / max(st.spec_issued, 1),
             "plies_per_rollout": float(st.rollout_plies) / max(st.rollouts, 1), "capped_rollout_frac": st.reason_hist[14] / max(st.rollouts, 1),
             "algorithmic_bytes_per_sim": bps, "roofline": roof}
        batch.close()
        logic.close()
        return r

    def guided_variant(sims=64):
        """Guided mode (the caller's network as nnet.predict) with a FREE evaluator: constant float32 priors and zero values resident in HBM,
        so that only the library's share of a round is timed (k_gmcts_step + k_gmcts_leaves); 65 536 games, device pointers."""
        from alphazeroforhnefatafl_amd import GuidedMCTS, MCTSArgs
        dev = torch.device("cuda:0")
        n, side = GAMES_PER_GPU, 11
        lg = BatchedGameLogic(abi.rules.COPENHAGEN, side, 128, device=0)
        A = lg.action_size
        bt = torch.zeros((n, side, side), dtype=torch.uint8, device=dev); stt = torch.zeros(n, dtype=torch.uint8, device=dev); wt = torch.zeros(n, dtype=torch.uint8, device=dev)
        pri = torch.rand((n, A), dtype=torch.float32, device=dev); val = torch.zeros(n, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()

        class Const:
            def predict_batch(self, *_):
                return pri.data_ptr(), val.data_ptr()
        b = lg.new_batch(n, abi.boards.COPENHAGEN)
        m = GuidedMCTS(b, Const(), MCTSArgs(numMCTSSims=sims, cpuct=1.0), edges_per_node=192, device=True, buffers=(bt.data_ptr(), stt.data_ptr(), wt.data_ptr()))
        m.search_all()                                           # first search allocates the arena
        b.reset_fen(abi.boards.COPENHAGEN, abi.rules.COPENHAGEN.starting_side)
        m.rounds = 0
        lg.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.search_all()
        lg.sync(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gs = b.gmcts_stats()
        r = {"workload": f"{n} x 11x11 guided MCTS (external evaluator), S={sims}, free evaluator (constant priors in HBM)", "value": gs.sims / dt, "unit": "sims/s",
             "rounds": m.rounds, "ms_per_round": 1e3 * dt / max(1, m.rounds), "predicts": int(gs.predicts), "faults": int(gs.faults)}
        b.close(); lg.close()
        del bt, stt, wt, pri, val
        torch.cuda.empty_cache()
        return r

    def selfplay_variant(sims=64, n_moves=8):
        """tafl_selfplay_run: every game runs n_moves searches of `sims` simulations, each followed by its most visited play, on the device and
        at its own pace - no barrier between the moves, so the games of the batch are at different phases of their searches and every launch
        finds the device full (the synchronous headline ends every search in a tail of nearly empty rounds).  Per game identical to the loop
        { tafl_mcts_run; tafl_mcts_play_best } (GPU test).  Positions: the games' own self-play from the start position, moves 0 .. n_moves-1."""
        logic = BatchedGameLogic(abi.rules.COPENHAGEN, 11, 128, device=0)
        batch = logic.new_batch(GAMES_PER_GPU, abi.boards.COPENHAGEN)
        batch.mcts_reserve(sims)
        batch.selfplay_run(2, sims, args.cpuct, args.seed, args.max_plies, want_plays=False)        # warm-up
        batch.reset_fen(abi.boards.COPENHAGEN, abi.rules.COPENHAGEN.starting_side)
        sync()
        t0 = time.perf_counter()
        batch.selfplay_run(n_moves, sims, args.cpuct, args.seed, args.max_plies, want_plays=False)
        sync()
        el = time.perf_counter() - t0
        st = batch.mcts_stats()
        assert st.faults == 0 and st.sims == GAMES_PER_GPU * sims * n_moves, (st.sims, st.faults)
        r = {"workload": f"{GAMES_PER_GPU} x 11x11 self-play on the device: {n_moves} moves per game, S={sims} per move, cap {args.max_plies}, every game at its own pace (tafl_selfplay_run)",
             "value": st.sims / el, "unit": "sims/s", "ms_per_move": el / n_moves * 1e3, "env_steps_per_sec": float(st.rollout_plies) / el,
             "spec_hit_rate": st.spec_hits / max(st.spec_issued, 1), "plies_per_rollout": float(st.rollout_plies) / max(st.rollouts, 1)}
        batch.close(); logic.close()
        return r

    out["selfplay_continuous_S64"] = selfplay_variant(64, 8)
    out["mcts_mixed_positions_S64"] = mcts_variant("copenhagen11", 64, 3, 1, mixed=True)
    out["mcts_S256"] = mcts_variant("copenhagen11", 256, 2, 1)
    out["mcts_S1000"] = mcts_variant("copenhagen11", 1000, 1, 1)
    out["mcts_13x13_S64"] = mcts_variant("copenhagen13", 64, 3, 1)
    out["mcts_brandubh7_S64"] = mcts_variant("brandubh7", 64, 3, 1)
    for board in ("copenhagen11", "copenhagen13"):
        rn, bn, side, wb = BOARDS[board]
        logic = BatchedGameLogic(getattr(abi.rules, rn), side, wb, device=0)
        for G in (4096, 65536):
            out[f"streamed_{board}_{G}"] = streamed_variant(logic, getattr(abi.boards, bn), side, G)
        logic.close()
    out["guided_engine_only_S64"] = guided_variant(64)
    return out


def make_digest(out):
    """<= 1 500 characters of the numbers a reader wants first, put LAST in the JSON line (a log tail keeps them)."""
    v = out.get("variants", {})

    def m(key):
        x = v.get(key)
        return "-" if not x else "%.1fM(hit %.2f)" % (x["value"] / 1e6, x.get("spec_hit_rate", 0.0))

    def st(key, k):
        x = v.get(key, {}).get(k)
        return "-" if not x else "%.1f" % x["us"]
    roof = out["roofline"]
    parts = ["S64 %.1fM sims/s %.1fms/step %.1fG plies/s" % (out["value"] / 1e6, out["ms_per_step"], out["env_steps_per_sec"] / 1e9),
             "hit %.2f" % (out["mcts"]["spec_hits"] / max(out["mcts"]["spec_issued"], 1)),
             "rollout launch %.2fms x%.0f/step union %.1fms overlap %.2f" % (roof["launch_avg_ms"], roof["launches_per_step"], roof["kernel_ms"], roof["overlap_factor"]),
             "tree %.3fms" % out["kernels_ms"]["k_mcts_tree"]["avg"],
             "self-play run (8 moves/game, no barrier between moves) " + m("selfplay_continuous_S64"), "mixed-positions S64 " + m("mcts_mixed_positions_S64"), "S256 " + m("mcts_S256"), "S1000 " + m("mcts_S1000"), "13x13 " + m("mcts_13x13_S64"),
             "brandubh7 " + m("mcts_brandubh7_S64"),
             "guided(free evaluator) %s" % ("-" if "guided_engine_only_S64" not in v else "%.1fM" % (v["guided_engine_only_S64"]["value"] / 1e6)),
             "streamed us @65536 11x11: counts %s masks %s step %s step_kth %s rollout %s" % tuple(st("streamed_copenhagen11_65536", k) for k in ("movegen_counts", "movegen_masks", "step", "step_kth", "rollout")),
             "@4096: masks %s step %s step_kth %s" % tuple(st("streamed_copenhagen11_4096", k) for k in ("movegen_masks", "step", "step_kth")),
             "13x13 @65536: masks %s step %s step_kth %s" % tuple(st("streamed_copenhagen13_65536", k) for k in ("movegen_masks", "step", "step_kth"))]
    cb = out.get("cpu_baseline")
    if cb:
        parts.append("cpu oracle %.0f sims/s x1 thread, %.0f x%d" % (cb["value"], cb["all_cores"]["value"], cb["all_cores"]["cores"]))
    return "; ".join(parts)[:1500]


# ----------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--sims", type=int, default=64, help="MCTS simulations per root (BASELINE config 3: 64/256/1000)")
    ap.add_argument("--games", type=int, default=GAMES_PER_GPU, help="concurrent games per GPU")
    ap.add_argument("--max-plies", type=int, default=512)
    ap.add_argument("--cpuct", type=float, default=1.0)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the `variants` object (S=256/1000, 13x13, streamed kernels)")
    ap.add_argument("--board", default="copenhagen11", choices=list(BOARDS),
                    help="copenhagen11 = the headline workload (BASELINE configs[2]); copenhagen13 = configs[4] (U256 multi-word path)")
    ap.add_argument("--pipeline", type=int, default=0, help="tuning: MCTS pipeline (0 default, 1 fused kernel, 2 two-kernel); results do not depend on it")
    ap.add_argument("--slots", type=int, default=0, help="tuning: playout slots per game the search is planned for (0 = from the batch size)")
    ap.add_argument("--parts", type=int, default=0, help="tuning: partitions of the batch on their own streams (0 = from the batch size)")
    ap.add_argument("--backend", default=None, choices=["auto", "nccl", "gloo"],
                    help="collective library for the timing barrier (the data path has no collective): auto = RCCL if it comes up on every rank, "
                         "else gloo (default); nccl = RCCL or exit non-zero; gloo (default with --single-device)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses GPU 0 (barrier over gloo)")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous check without a GPU: every rank joins the gloo group, "
                    "reports its device and game-id shard, rank 0 prints the table; no engine call is made")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    backend = args.backend or ("gloo" if args.single_device else "auto")

    has_env = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not has_env:
        # invoked directly: become the launcher.  Nothing has touched torch / HIP in this process.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    from alphazeroforhnefatafl_amd import dist as tdist
    rank, local_rank, world = tdist.env_rank_world()
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}: launch one rank per GPU "
                         f"(python bench.py --gpus N, or torch.distributed.run --nproc-per-node N bench.py --gpus N)")

    import torch
    import torch.distributed as dist
    if args.dry_run:
        tdist.init("gloo", rank, world)
        tdist.barrier(world)
        bases = tdist.gather_over_ranks(float(tdist.shard_base(rank, args.games)), world)
        devs = tdist.gather_over_ranks(float(0 if args.single_device else local_rank), world)
        slow = tdist.max_over_ranks(1.0 + rank, world)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "shard_bases": [int(b) for b in bases], "devices": [int(d) for d in devs],
                              "max_over_ranks": slow, "games_total": args.games * world,
                              "launcher": "bench.py child processes" if os.environ.get("TAFL_BENCH_CHILD") else ("torch.distributed.run" if has_env else "single process")}), flush=True)
        if world > 1:
            tdist.barrier(world)
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    if args.single_device:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: GPU {local_rank} does not exist ({torch.cuda.device_count()} visible); use --single-device to rehearse on one GPU")
    torch.cuda.set_device(local_rank)
    tdist.init(backend, rank, world)    # RCCL: only for the barrier; the reductions of the elapsed time / counters are host scalars over gloo
    red_dev = "cpu"

    from alphazeroforhnefatafl_amd import abi
    from alphazeroforhnefatafl_amd.engine import BatchedGameLogic

    stream = torch.cuda.Stream()
    rn, bn, side_, wb_ = BOARDS[args.board]
    rules_, fen_ = getattr(abi.rules, rn), getattr(abi.boards, bn)
    logic = BatchedGameLogic(rules_, side_, wb_, device=local_rank, stream=stream.cuda_stream)
    G = args.games
    batch = logic.new_batch(G, fen_)                          # synthetic data: every game at the start position
    batch.mcts_reserve(args.sims)
    base = tdist.shard_base(rank, G)                          # contiguous global game-id shards, no collective on the data path

    def barrier():
        torch.cuda.synchronize()
        tdist.barrier(world)
        torch.cuda.synchronize()

    my_elapsed, stats, kt = timed_mcts(logic, batch, args.sims, args.cpuct, args.seed, args.max_plies, base, args.steps, args.warmup, barrier,
                                       flags=abi.mcts_tune(args.pipeline, args.slots, args.parts))
    if stats.sims != G * args.sims or stats.faults != 0:
        raise SystemExit(f"rank {rank}: the last step ran {stats.sims} simulations with {stats.faults} faults, expected {G * args.sims} / 0")
    elapsed = tdist.max_over_ranks(my_elapsed, world, device=red_dev)
    per_rank = tdist.gather_over_ranks(my_elapsed, world, device=red_dev)
    total_sims = tdist.sum_over_ranks(float(stats.sims), world, device=red_dev) * args.steps
    total_plies = tdist.sum_over_ranks(float(stats.rollout_plies), world, device=red_dev) * args.steps
    if rank == 0:
        headline = G == GAMES_PER_GPU and args.board == "copenhagen11" and args.sims == 64
        traffic = tsrc = None
        probe_kname = "k_mcts_fused" if kt["tree"][1] == 0 else "k_mcts_rollout"
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % probe_kname)
        if headline and os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            traffic = tj.get("hbm_bytes_per_step")
            tsrc = "profiles/traffic_%s.json (committed rocprofv3 PMC passes of this command, not this run): %s" % (probe_kname, tj.get("source", ""))
        roof, kname, bps, d_bar, c_bar = mcts_roofline(stats, kt, args.steps, side_, traffic, tsrc)
        roof["note"] = ("register-resident playouts (S_g + 4 algorithmic bytes each) + the tree phase's node / edge records: by construction far from "
                        "the HBM roof; the binding limit is integer VALU issue (DESIGN.md section 6: per-class issue costs in profiles/r01_valu_rates, "
                        "opcode-class histogram in profiles/r02_opclass)")
        tree_ms, tree_n = kt["tree"]
        roll_ms, roll_n = kt["rollout"]
        out = {
            "metric": "mcts_sims_per_sec", "value": total_sims / elapsed, "unit": "sims/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[2]" if headline else "variant") +
                                   f": full MCTS (select/expand/random-rollout/backup), {G} concurrent {side_}x{side_} "
                                   f"{args.board} games per GPU from the start position" + (f"; x{world} GPUs = configs[3]" if headline and world == 8 else ""),
                       "games_per_gpu": G, "games_total": G * world, "sims_per_root": args.sims, "max_rollout_plies": args.max_plies,
                       "c_puct": args.cpuct, "seed": args.seed, "pipeline": args.pipeline, "slots": args.slots, "sharding": f"game-id ranges x{world}, no collectives",
                       "launcher": "bench.py child processes" if os.environ.get("TAFL_BENCH_CHILD") else ("torch.distributed.run" if has_env else "single process"),
                       "barrier_backend": tdist.backend_in_use()[0], "barrier_note": tdist.backend_in_use()[1],
                       "single_device_rehearsal": bool(args.single_device)},
            "per_rank_sims_per_sec": [G * args.sims * args.steps / t for t in per_rank],
            "env_steps_per_sec": total_plies / elapsed,
            "plies_per_rollout": float(stats.rollout_plies) / max(stats.rollouts, 1),
            "mcts": {"sims": int(stats.sims), "rollouts": int(stats.rollouts), "terminal_hits": int(stats.terminal_hits),
                     "mean_select_depth": d_bar, "mean_children_scanned": c_bar, "faults": int(stats.faults), "spec_issued": int(stats.spec_issued), "spec_hits": int(stats.spec_hits),
                     "reason_hist": [int(x) for x in stats.reason_hist],
                     "algorithmic_bytes_per_sim": bps,
                     "hbm_frac_sims": (total_sims / elapsed / world) * bps / (HBM_PEAK_GBS * 1e9)},
            "kernels_ms": {kname: {"avg": roll_ms / max(roll_n, 1), "launches": int(roll_n), "total_per_step": roll_ms / args.steps},
                           "k_mcts_tree": {"avg": tree_ms / max(tree_n, 1), "launches": int(tree_n), "total_per_step": tree_ms / args.steps}},
            "roofline": roof,
        }
        # the honest ceiling of this path is VALU issue, not HBM: wave64 VALU instructions per step from the committed SQ counter pass
        # of this command (profiles/, like `traffic`), over this run's measured step time
        ppath = os.path.join(ROOT, "profiles", "valu_k_mcts_rollout.json")
        if headline and os.path.exists(ppath):
            with open(ppath) as f:
                pj = json.load(f)
            insts = float(pj.get("valu_wave_instructions_per_step", 0.0))
            if insts > 0:
                ach = insts / (elapsed / args.steps)
                out["roofline_valu"] = {"bound": "valu_issue", "achieved": ach, "peak": VALU_ISSUE_PEAK, "unit": "wave64 VALU instructions/s", "frac": ach / VALU_ISSUE_PEAK,
                                        "instructions_per_step": insts, "source": "profiles/valu_k_mcts_rollout.json: " + pj.get("source", "")}
        batch.close()
        if world == 1 and not args.no_variants and headline:
            out["variants"] = run_variants(args, torch)
        if world == 1 and not args.no_cpu_baseline and args.board == "copenhagen11":
            out["cpu_baseline"] = cpu_baseline(args.sims, args.cpuct, args.seed, args.max_plies)
        out["digest"] = make_digest(out)                     # last key on purpose
        print(json.dumps(out), flush=True)
    if world > 1:
        tdist.barrier(world)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
