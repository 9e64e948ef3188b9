#!/usr/bin/env python3
"""Self-play throughput of the whole loop on one MI355X: repeat { tafl_mcts_run (S sims per root); tafl_mcts_play_best } until
every game is over or `--max-moves` is reached, nothing leaving the device in between.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=65536)
    ap.add_argument("--sims", type=int, default=64)
    ap.add_argument("--max-moves", type=int, default=40)
    ap.add_argument("--board", default="copenhagen11", choices=["copenhagen11", "brandubh7"])
    a = ap.parse_args()
    from alphazeroforhnefatafl_amd import BatchedGameLogic, abi, boards, rules
    if a.board == "copenhagen11":
        lg, fen = BatchedGameLogic(rules.COPENHAGEN, 11), boards.COPENHAGEN
    else:
        lg, fen = BatchedGameLogic(rules.BRANDUBH, 7), boards.BRANDUBH
    b = lg.new_batch(a.games, fen)
    b.mcts_run(a.sims, 1.0, 1, 512)                       # warm-up: arena allocation, code load
    b.reset_fen(fen, abi.ATTACKER)
    lg.sync()
    t0 = time.perf_counter()
    moves = 0
    for m in range(a.max_moves):
        b.mcts_run(a.sims, 1.0, 1000 + m, 512)
        b.mcts_play_best(want_results=False)
        moves += 1
    lg.sync()
    dt = time.perf_counter() - t0
    st = b.download()
    over = sum(1 for g in range(a.games) if st[g].status != abi.ONGOING)
    print(json.dumps({"board": a.board, "games": a.games, "sims_per_move": a.sims, "moves_played": moves, "seconds": round(dt, 3),
                      "game_moves_per_sec": a.games * moves / dt, "mcts_sims_per_sec": a.games * moves * a.sims / dt, "games_over": over}))


if __name__ == "__main__":
    main()
