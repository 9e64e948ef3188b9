#!/usr/bin/env python3
"""Per-round work of one MCTS search on the GPU (two-kernel pipeline): playouts requested / run in every round of the first half of the
batch, next to the step time.  Measurement aid for the slot-issue policy (tools/spec_model.py is its CPU-side counterpart)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazeroforhnefatafl_amd import abi
from alphazeroforhnefatafl_amd.engine import BatchedGameLogic


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--board", default="copenhagen11"); ap.add_argument("--sims", type=int, default=64)
    ap.add_argument("--games", type=int, default=65536); ap.add_argument("--slots", type=int, default=0); ap.add_argument("--cap", type=int, default=512); ap.add_argument("--parts", type=int, default=0)
    a = ap.parse_args()
    rules, fen, n, wb = {"copenhagen11": (abi.rules.COPENHAGEN, abi.boards.COPENHAGEN, 11, 128), "copenhagen13": (abi.rules.COPENHAGEN, abi.boards.COPENHAGEN13, 13, 256),
                         "brandubh7": (abi.rules.BRANDUBH, abi.boards.BRANDUBH, 7, 64)}[a.board]
    lg = BatchedGameLogic(rules, n, wb)
    b = lg.new_batch(a.games, fen)
    fl = abi.mcts_tune(0, a.slots, a.parts)
    b.mcts_run(a.sims, 1.0, 2, a.cap, flags=fl)
    lg.sync()
    t0 = time.perf_counter()
    b.mcts_run(a.sims, 1.0, 2, a.cap, flags=fl)
    lg.sync()
    dt = time.perf_counter() - t0
    req, run = b.mcts_round_trace()
    st = b.mcts_stats()
    print(json.dumps({"board": a.board, "sims": a.sims, "games": a.games, "slots": a.slots, "ms": dt * 1e3, "Msims_per_s": a.games * a.sims / dt / 1e6,
                      "rounds": len(req), "spec_hits": st.spec_hits, "spec_issued": st.spec_issued, "requested_first_half": req, "run_first_half": run}))


if __name__ == "__main__":
    main()
