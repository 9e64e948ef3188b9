#!/usr/bin/env python3
"""Generates tests/golden/mcts_golden.json by running the REFERENCE's own src/mcts.py.

Runs only in the build container (needs /root/reference; nothing of the reference travels:
the output is data — inputs and expected root statistics).  The reference `MCTS` class
(src/mcts.py:11-136) is driven through the alpha-zero-general duck-typed `game`/`nnet`
protocol it expects (src/mcts.py:40-41,75,78,85-86,122-123) by a thin adaptor whose game
methods call OUR CPU oracle and whose `predict` returns an all-ones prior plus the value of one
seeded random playout (SURVEY.md §8a resolution, DESIGN.md "MCTS semantics"); the playout's RNG
stream is a function of (seed, game id, position) - `logic.state_hash`, include/taflhip.h "leaf
key" - as a network's predict is a function of the board.  States are keyed by their move path,
so no transpositions merge (the tree is explicit, as in src/mcts.rs).

Usage:  python tests/golden/make_mcts_golden.py
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")
sys.setrecursionlimit(20000)

import numpy as np  # noqa: E402

import mcts as ref_mcts  # noqa: E402  the reference's src/mcts.py

from alphazeroforhnefatafl_amd import abi  # noqa: E402
from oracle import oracle as orc  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
from stub_net import matrix_bytes_of, stub_predict  # noqa: E402

DRAW_VALUE = 1e-4   # alpha-zero-general convention for a drawn game (non-zero = ended)


class Board:
    __slots__ = ("state", "path")

    def __init__(self, state, path):
        self.state = state
        self.path = path


class TaflGame:
    def __init__(self, logic, side_len):
        self.logic = logic
        self.n = side_len
        self.root_calls = 0

    def getActionSize(self):
        return abi.action_size(self.n)

    def getValidMoves(self, b, player):
        v = np.zeros(self.getActionSize(), dtype=np.float64)
        for p in self.logic.all_plays(b.state):
            v[abi.action_encode(self.n, p)] = 1
        return v

    def getNextState(self, b, player, a):
        new, _ = self.logic.do_valid_play(abi.action_decode(self.n, int(a)), b.state)
        return Board(new, b.path + (int(a),)), -player

    def getCanonicalForm(self, b, player):
        return b            # the state carries its side to move

    def getGameEnded(self, b, player):
        status, _reason, winner = b.state.status
        if status == abi.ONGOING:
            return 0
        if status == abi.DRAW:
            return DRAW_VALUE
        return 1.0 if winner == b.state.side_to_play else -1.0

    def stringRepresentation(self, b):
        if not b.path:
            self.root_calls += 1      # one call per search(root) + one in getActionProb
        return ",".join(map(str, b.path))


class RolloutNet:
    def __init__(self, game, seed, game_id, max_plies):
        self.game, self.seed, self.game_id, self.max_plies = game, seed, game_id, max_plies

    def predict(self, b):
        sim = self.game.logic.state_hash(b.state)          # the playout is a function of (seed, game, board): as a network's predict is
        r = self.game.logic.rollout(b.state, self.seed, self.game_id, sim, self.max_plies)
        return np.ones(self.game.getActionSize(), dtype=np.float64), float(r.value)


class StubNet:
    """nnet.predict = tests/stub_net.py on the board_to_matrix planes; the value is handed over as a Python float."""

    def __init__(self, game, salt):
        self.game, self.salt = game, salt
        self.calls = 0

    def predict(self, b):
        self.calls += 1
        pri, v = stub_predict(matrix_bytes_of(b.state.board_to_matrix()), int(b.state.side_to_play), self.game.getActionSize(), self.salt)
        return pri, float(v)


class Args:
    def __init__(self, n_sims, cpuct):
        self.numMCTSSims = n_sims
        self.cpuct = cpuct


CASES = [
    dict(name="config1_brandubh_1000", rules="brandubh", fen=abi.boards.BRANDUBH, side="starting", n_sims=1000,
         cpuct=1.0, seed=0, game_id=0, max_plies=256),
    dict(name="copenhagen_start_200", rules="copenhagen", fen=abi.boards.COPENHAGEN, side="starting", n_sims=200,
         cpuct=1.0, seed=2, game_id=0, max_plies=512),
    dict(name="copenhagen_start_64_id5", rules="copenhagen", fen=abi.boards.COPENHAGEN, side="starting", n_sims=64,
         cpuct=1.5, seed=2, game_id=5, max_plies=512),
    dict(name="copenhagen_midgame_300", rules="copenhagen", fen=None, advance=dict(seed=1, game_id=37, plies=37),
         side="starting", n_sims=300, cpuct=1.0, seed=7, game_id=37, max_plies=512),
    dict(name="tablut_start_300", rules="tablut", fen=abi.boards.TABLUT, side="starting", n_sims=300, cpuct=1.0,
         seed=3, game_id=1, max_plies=300),
    dict(name="brandubh_near_escape_150", rules="brandubh", fen="7/7/3t3/2t4/7/5K1/3t3", side="D", n_sims=150,
         cpuct=1.0, seed=4, game_id=2, max_plies=128),
    dict(name="brandubh_king_threat_150", rules="brandubh", fen="3t3/7/7/t3K1t/7/7/3t3", side="A", n_sims=150,
         cpuct=2.0, seed=5, game_id=3, max_plies=128),
    dict(name="copenhagen13_start_48", rules="copenhagen", fen=abi.boards.COPENHAGEN13, side="starting", n_sims=48,
         cpuct=1.0, seed=9, game_id=0, max_plies=256),
]


GUIDED_CASES = [
    dict(name="guided_brandubh_400", rules="brandubh", fen=abi.boards.BRANDUBH, side="starting", n_sims=400, cpuct=1.0, salt=1),
    dict(name="guided_copenhagen_start_150", rules="copenhagen", fen=abi.boards.COPENHAGEN, side="starting", n_sims=150, cpuct=1.0, salt=2),
    dict(name="guided_copenhagen_midgame_250", rules="copenhagen", fen=None, advance=dict(seed=1, game_id=37, plies=37), side="starting",
         n_sims=250, cpuct=2.5, salt=3),
    dict(name="guided_tablut_200", rules="tablut", fen=abi.boards.TABLUT, side="starting", n_sims=200, cpuct=0.5, salt=4),
    dict(name="guided_brandubh_near_escape_120", rules="brandubh", fen="7/7/3t3/2t4/7/5K1/3t3", side="D", n_sims=120, cpuct=1.0, salt=5),
    dict(name="guided_copenhagen13_60", rules="copenhagen", fen=abi.boards.COPENHAGEN13, side="starting", n_sims=60, cpuct=1.0, salt=6),
    dict(name="guided_masked_root_90", rules="brandubh", fen=abi.boards.BRANDUBH, side="starting", n_sims=90, cpuct=1.0, salt=None),
]


def run_guided_case(c):
    rules = abi.rules.BY_NAME[c["rules"]]
    side = rules.starting_side if c["side"] == "starting" else (abi.ATTACKER if c["side"] == "A" else abi.DEFENDER)
    fen = c["fen"] or abi.boards.COPENHAGEN
    n = abi.fen_side_len(fen)
    wb = abi.word_bits_for(n)
    logic = orc.GameLogic(rules, n)
    st = orc.GameState(fen, side, wb)
    if c.get("advance"):
        a = c["advance"]
        st = logic.random_advance(st, a["seed"], a["game_id"], a["plies"])
    game = TaflGame(logic, n)
    salt = c["salt"]
    if salt is None:      # a salt for which the ROOT gets all-zero priors: the workaround branch at the root (mcts.py:91-98)
        mb = matrix_bytes_of(st.board_to_matrix())
        salt = next(x for x in range(256) if not stub_predict(mb, int(st.side_to_play), game.getActionSize(), x)[0].any())
    net = StubNet(game, salt)
    m = ref_mcts.MCTS(game, net, Args(c["n_sims"], c["cpuct"]))
    root = Board(st, ())
    probs = m.getActionProb(root, temp=1)
    s = game.stringRepresentation(root)
    children = [[a, int(m.Nsa[(s, a)]), float(m.Qsa[(s, a)]).hex()] for a in range(game.getActionSize()) if (s, a) in m.Nsa]
    out = dict(c)
    out.update(salt=salt, fen=st.to_fen(), side_to_play=int(st.side_to_play), state_hex=bytes(st.to_abi()).hex(), word_bits=wb, side_len=n,
               root_ns=int(m.Ns[s]), root_children=children, n_tree_states=len(m.Ps), predict_calls=net.calls,
               n_terminal_states=sum(1 for v in m.Es.values() if v != 0),
               root_priors_nonzero=[(i, float(p).hex()) for i, p in enumerate(m.Ps[s]) if p != 0],
               probs_temp1_nonzero=[(i, float(p).hex()) for i, p in enumerate(probs) if p != 0])
    return out


def run_case(c):
    rules = abi.rules.BY_NAME[c["rules"]]
    side = rules.starting_side if c["side"] == "starting" else (abi.ATTACKER if c["side"] == "A" else abi.DEFENDER)
    fen = c["fen"] or abi.boards.COPENHAGEN
    n = abi.fen_side_len(fen)
    wb = abi.word_bits_for(n)
    logic = orc.GameLogic(rules, n)
    st = orc.GameState(fen, side, wb)
    if c.get("advance"):
        a = c["advance"]
        st = logic.random_advance(st, a["seed"], a["game_id"], a["plies"])
    game = TaflGame(logic, n)
    net = RolloutNet(game, c["seed"], c["game_id"], c["max_plies"])
    m = ref_mcts.MCTS(game, net, Args(c["n_sims"], c["cpuct"]))
    root = Board(st, ())
    probs = m.getActionProb(root, temp=1)
    s = game.stringRepresentation(root)
    children = []
    for a in range(game.getActionSize()):
        if (s, a) in m.Nsa:
            children.append([a, int(m.Nsa[(s, a)]), float(m.Qsa[(s, a)]).hex()])
    out = dict(c)
    out["fen"] = st.to_fen()
    out["side_to_play"] = int(st.side_to_play)
    out["state_hex"] = bytes(st.to_abi()).hex()
    out["word_bits"] = wb
    out["side_len"] = n
    out["root_ns"] = int(m.Ns[s])
    out["root_children"] = children
    out["n_tree_states"] = len(m.Ps)
    out["n_terminal_states"] = sum(1 for v in m.Es.values() if v != 0)
    nz = [(i, float(p).hex()) for i, p in enumerate(probs) if p != 0]
    out["probs_temp1_nonzero"] = nz
    return out


def main():
    res = dict(_comment="Generated by tests/golden/make_mcts_golden.py from the reference's src/mcts.py "
                        "(MCTS.getActionProb/search) over the oracle adaptor. Qsa/probs are float.hex().",
               draw_value=DRAW_VALUE, cases=[run_case(c) for c in CASES],
               guided_comment="guided_cases: the same reference MCTS with nnet.predict = tests/stub_net.py (non-uniform float32 priors, "
                              "value passed as a Python float); root_priors_nonzero = Ps[root] after masking / renormalising.",
               guided_cases=[run_guided_case(c) for c in GUIDED_CASES])
    with open(os.path.join(HERE, "mcts_golden.json"), "w") as f:
        json.dump(res, f, indent=1)
    for c in res["guided_cases"]:
        print(c["name"], "Ns", c["root_ns"], "children", len(c["root_children"]), "states", c["n_tree_states"], "terminal", c["n_terminal_states"])
    for c in res["cases"]:
        print(c["name"], "Ns", c["root_ns"], "children", len(c["root_children"]), "states", c["n_tree_states"],
              "terminal", c["n_terminal_states"])


if __name__ == "__main__":
    main()
