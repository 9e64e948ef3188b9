"""Guided MCTS (external evaluator) — the device code of tafl_guided.hpp compiled for the host, against (i) the vectors the
reference's own mcts.py produced with the stub network and (ii) the literal oracle on random batches.  CPU only."""
import json
import os

import pytest

from alphazeroforhnefatafl_amd import abi
from oracle import oracle as orc
from tests import guided_util as gu
from tests import parity_util as pu
from tests.hostsim import hostsim
from tests.stub_net import matrix_bytes_of, stub_predict

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "mcts_golden.json")) as f:
    GOLD = json.load(f)


@pytest.mark.parametrize("case", GOLD["guided_cases"], ids=[c["name"] for c in GOLD["guided_cases"]])
def test_engine_guided_matches_reference_mcts_py(case):
    n, wb = case["side_len"], case["word_bits"]
    hs = hostsim.HostSim(abi.rules.BY_NAME[case["rules"]], n, wb)
    st = abi.TaflState.from_buffer_copy(bytes.fromhex(case["state_hex"]))
    states = (abi.TaflState * 1)(st)
    kids, counts, rounds = gu.run_hostsim_guided(hs, hostsim.lib(), states, 1, case["n_sims"], case["cpuct"], [case["salt"]])
    assert [[a, v, q] for a, v, q in kids[0]] == case["root_children"]
    assert counts[0] == case["n_sims"] and counts[1] == case["predict_calls"] and counts[3] == 0
    assert rounds == case["predict_calls"]


@pytest.mark.parametrize("cfg", ["brandubh7", "copenhagen11", "copenhagen13", "tablut9", "magpie7"])
def test_engine_guided_matches_oracle_on_a_batch(cfg):
    rules, fen, wb = pu.CONFIGS[cfg]
    n = abi.fen_side_len(fen)
    G, S = 12, 40
    lg = orc.GameLogic(rules, n)
    base = orc.GameState(fen, rules.starting_side, wb)
    states = (abi.TaflState * G)(*[lg.random_advance(base, 21, g, (g * 7) % 60).to_abi() for g in range(G)])
    salts = [(3 * g + 1) % 256 for g in range(G)]
    hs = hostsim.HostSim(rules, n, wb)
    kids, counts, _ = gu.run_hostsim_guided(hs, hostsim.lib(), states, G, S, 1.25, salts)
    A = abi.action_size(n)
    for g in range(G):
        gs = orc.GameState.from_abi(states[g], wb)
        ok, root_ns, _pri, _cnt = lg.gmcts(gs, S, 1.25, lambda s, g=g: stub_predict(matrix_bytes_of(s.board_to_matrix()), int(s.side_to_play), A, salts[g]), wb)
        assert [(a, v, float(q).hex()) for (_p, a, v, q) in ok] == kids[g], f"game {g}"
    assert counts[0] == G * S and counts[3] == 0


def test_arena_overflow_raises_the_fault_flag():
    rules, fen, wb = pu.CONFIGS["copenhagen11"]
    n = abi.fen_side_len(fen)
    hs = hostsim.HostSim(rules, n, wb)
    st = orc.GameState(fen, rules.starting_side, wb).to_abi()
    states = (abi.TaflState * 1)(st)
    kids, counts, _ = gu.run_hostsim_guided(hs, hostsim.lib(), states, 1, 30, 1.0, [2], edges_per_node=40)   # 116 legal plays at the root
    assert counts[3] >= 1 and counts[0] < 30
