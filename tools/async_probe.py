#!/usr/bin/env python3
"""Measurement: one synchronous search of 65 536 games per step vs two half-batches of 32 768 searched asynchronously half a search apart
(tafl_mcts_run_async / _after / tafl_mcts_wait), for a few (parts, slots) settings.  Prints one JSON object per setting."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alphazeroforhnefatafl_amd import abi, _lib
if os.environ.get('PROBE_LIB'):
    _lib.LIB_PATH = os.environ['PROBE_LIB']          # measurement only: A/B another build of the library
from alphazeroforhnefatafl_amd.engine import BatchedGameLogic

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
CAP, G = 512, 65536
logic = BatchedGameLogic(abi.rules.COPENHAGEN, 11, 128, device=0)


def sync_mode(parts, slots):
    b = logic.new_batch(G, abi.boards.COPENHAGEN)
    fl = abi.mcts_tune(0, slots, parts)
    for _ in range(2):
        b.mcts_run(S, 1.0, 2, CAP, 0, flags=fl)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(STEPS):
        b.mcts_run(S, 1.0, 2, CAP, 0, flags=fl)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = b.mcts_stats(); assert st.sims == G * S and st.faults == 0
    b.close()
    return {"mode": "sync", "parts": parts, "slots": slots, "sims_per_s": G * S * STEPS / dt, "ms_per_step": dt / STEPS * 1e3, "hit": st.spec_hits / max(st.spec_issued, 1)}


def async_mode(nb, parts, slots, share=0):
    H = G // nb // 64 * 64
    bs = [logic.new_batch(H, abi.boards.COPENHAGEN) for _ in range(nb)]
    fl = abi.mcts_tune(0, slots, parts, share)

    def go(steps):
        for i, b in enumerate(bs):
            b.mcts_run_async(S, 1.0, 2, CAP, i * H, flags=fl, after=bs[i - 1] if i else None)
        for _ in range(steps - 1):
            for i, b in enumerate(bs):
                b.mcts_wait()
                b.mcts_run_async(S, 1.0, 2, CAP, i * H, flags=fl)
        for b in bs:
            b.mcts_wait()
    go(2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    go(STEPS)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    hit = 0.0
    for b in bs:
        st = b.mcts_stats(); assert st.sims == H * S and st.faults == 0; hit += st.spec_hits / max(st.spec_issued, 1) / nb
    for b in bs:
        b.close()
    return {"mode": "async x%d" % nb, "parts": parts, "slots": slots, "share": share, "sims_per_s": H * nb * S * STEPS / dt, "ms_per_step": dt / STEPS * 1e3, "hit": hit}


MODES = sys.argv[3].split(",") if len(sys.argv) > 3 else ["sync", "async"]
if "lone" in MODES:          # one half-size batch searched alone, synchronously
    for parts in (1, 2):
        b = logic.new_batch(G // 2, abi.boards.COPENHAGEN)
        fl = abi.mcts_tune(0, 4, parts)
        b.mcts_run(S, 1.0, 2, CAP, 0, flags=fl)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(STEPS):
            b.mcts_run(S, 1.0, 2, CAP, 0, flags=fl)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        req, run = b.mcts_round_trace()
        print(json.dumps({"mode": "lone half batch", "parts": parts, "ms_per_search": dt / STEPS * 1e3, "rounds_run": run}), flush=True)
        b.close()
if "sync" in MODES:
    for parts, slots in ((2, 0), (2, 4)):
        print(json.dumps(sync_mode(parts, slots)), flush=True)
if "async" in MODES:
    for nb, parts, slots, share in ((2, 1, 4, 0), (2, 1, 4, 2), (2, 1, 0, 2), (2, 1, 5, 2)):
        print(json.dumps(async_mode(nb, parts, slots, share)), flush=True)
if "async4" in MODES:
    for nb, parts, slots, share in ((4, 1, 4, 4), (4, 1, 0, 4), (3, 1, 4, 3), (2, 2, 4, 2)):
        if G % nb == 0 or nb == 3:
            print(json.dumps(async_mode(nb, parts, slots, share)), flush=True)
if "async1" in MODES:
    print(json.dumps(async_mode(2, 1, 4)), flush=True)
