#!/usr/bin/env python3
"""Cost model of the MCTS simulation pipeline from a HOST run of the product's per-game code (tests/hostsim): rounds, playouts per
round, prediction hit rate, and the step time the per-round work predicts on an MI355X using the measured playout-round times of
tools/rollout_bench.hip (11x11, cap 512: 1.27 / 1.89 / 2.51 / 3.21 ms at 1 / 2 / 3 / 4 waves per SIMD).  Development tool for the
slot-issue policy (how many predicted simulations a game runs beside the pending one); no GPU needed.

  python tools/spec_model.py --board copenhagen11 --sims 64 --games 96 --slots 8 --target 4
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazeroforhnefatafl_amd import abi
from alphazeroforhnefatafl_amd.abi import TaflMctsParams
from tests.hostsim import hostsim
from tests.hostsim.hostsim import HostSim

ROUND_MS = {"copenhagen11": [0.0, 1.27, 1.89, 2.51, 3.21], "copenhagen13": [0.0, 1.86, 2.80, 3.86, 4.92], "brandubh7": [0.0, 1.07, 1.29, 1.45, 1.84]}


def t_round(board, waves):
    tab = ROUND_MS[board]
    if waves >= 4:
        return tab[4] * waves / 4.0
    if waves <= 0:
        return 0.0
    if waves < 1:                                 # a lone wave on a SIMD still takes a full playout
        return tab[1]
    i = int(waves)
    return tab[i] + (tab[i + 1] - tab[i]) * (waves - i)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--board", default="copenhagen11")
    ap.add_argument("--sims", type=int, default=64)
    ap.add_argument("--games", type=int, default=96)
    ap.add_argument("--slots", type=int, default=8)
    ap.add_argument("--target", type=int, default=4)
    ap.add_argument("--cap", type=int, default=512)
    ap.add_argument("--capacity", type=float, default=0.0, help="playouts per round the device holds, in units of games (4 = 4 waves per SIMD at 65 536 games); 0 = unlimited")
    ap.add_argument("--advance", type=int, default=0, help="random plies played before the search (mid-game positions)")
    a = ap.parse_args()
    rules, fen, n, wb = {"copenhagen11": (abi.rules.COPENHAGEN, abi.boards.COPENHAGEN, 11, 128), "copenhagen13": (abi.rules.COPENHAGEN, abi.boards.COPENHAGEN13, 13, 256),
                         "brandubh7": (abi.rules.BRANDUBH, abi.boards.BRANDUBH, 7, 64)}[a.board]
    hs = HostSim(rules, n, wb)
    from oracle import oracle as orc
    st0 = orc.GameState(fen, rules.starting_side, wb).to_abi()
    states = (abi.TaflState * a.games)(*[st0] * a.games)
    if a.advance:
        import ctypes as C
        hs.random_advance(states, a.games, 5, (C.c_uint32 * a.games)(*[a.advance] * a.games), 0)
    hostsim.set_spec_k(a.slots, a.target, int(a.capacity * a.games))
    p = TaflMctsParams(a.sims, a.cap, 1.0, 2, 0, 0)
    kids, cnt, st = hs.mcts(states, a.games, p, 0)
    work = hostsim.round_work()
    scale = 65536.0 / a.games / 65536.0           # playouts per round -> waves per SIMD at 65 536 games
    ms = sum(t_round(a.board, w * scale) for w in work)
    executed = sum(work)
    print(f"{a.board} S={a.sims} slots={a.slots} target={a.target}: rounds {len(work)}, playouts executed {executed} for {st.rollouts} consumed "
          f"(waste {executed / max(st.rollouts, 1) - 1:.3f}), predicted {st.spec_hits}/{st.spec_issued} = {st.spec_hits / max(st.spec_issued, 1):.3f}")
    print(f"  modelled playout time per step at 65 536 games: {ms:.1f} ms -> {65536 * a.sims / ms / 1e3:.1f} M sims/s (tree phase not included)")
    print("  waves/SIMD per round:", " ".join(f"{w * scale:.2f}" for w in work[:40]), "..." if len(work) > 40 else "")


if __name__ == "__main__":
    main()
