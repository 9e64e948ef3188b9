#!/usr/bin/env python3
"""Measurement: tafl_selfplay_run (every game at its own pace) against the synchronous loop of searches and plays; per-round playout counts
and kernel timings of the self-play run."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alphazeroforhnefatafl_amd import abi
from alphazeroforhnefatafl_amd.engine import BatchedGameLogic, KC_MCTS_ROLLOUT, KC_MCTS_TREE
S, MOVES, G, CAP = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 8, 65536, 512
logic = BatchedGameLogic(abi.rules.COPENHAGEN, 11, 128, device=0)
b = logic.new_batch(G, abi.boards.COPENHAGEN)
b.selfplay_run(2, S, 1.0, 2, CAP, want_plays=False)
b.reset_fen(abi.boards.COPENHAGEN, 0)
logic.timing_reset(); logic.timing_enable(True)
torch.cuda.synchronize(); t0 = time.perf_counter()
b.selfplay_run(MOVES, S, 1.0, 2, CAP, want_plays=False)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
logic.timing_enable(False)
st = b.mcts_stats()
req, run = b.mcts_round_trace()
rm, rn = logic.timing_get(KC_MCTS_ROLLOUT); tm, tn = logic.timing_get(KC_MCTS_TREE)
ru, _ = logic.timing_get_union(KC_MCTS_ROLLOUT)
print(json.dumps({"mode": "selfplay", "moves": MOVES, "ms_per_move": dt / MOVES * 1e3, "sims_per_s": st.sims / dt, "rounds": len(run), "rounds_per_move": len(run) / MOVES,
                  "rollout_avg_ms": rm / rn, "tree_avg_ms": tm / tn, "rollout_union_ms": ru, "hit": st.spec_hits / max(st.spec_issued, 1),
                  "requested_first_half_batch": req[:60], "run": run[:60]}))
b.reset_fen(abi.boards.COPENHAGEN, 0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for m in range(MOVES):
    b.mcts_run(S, 1.0, 2, CAP, 0, sim_offset=m * S)
    b.mcts_play_best(want_results=False)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"mode": "synchronous loop", "moves": MOVES, "ms_per_move": dt / MOVES * 1e3, "sims_per_s": G * S * MOVES / dt}))
