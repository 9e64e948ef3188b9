"""Pins the C restatement of the MCTS arithmetic (oracle/tafl_oracle.c mcts_search) against
golden vectors produced by the reference's own src/mcts.py (tests/golden/make_mcts_golden.py).
Bit-exact: visit counts, Ns[root] and Qsa as float64 bit patterns.  CPU only.
"""
import ctypes as C
import json
import os

import pytest

from alphazeroforhnefatafl_amd import abi
from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "mcts_golden.json")) as f:
    GOLD = json.load(f)


def state_from_case(c):
    st = abi.TaflState.from_buffer_copy(bytes.fromhex(c["state_hex"]))
    return orc.GameState.from_abi(st, c["word_bits"])


@pytest.mark.parametrize("case", GOLD["cases"], ids=[c["name"] for c in GOLD["cases"]])
def test_c_mcts_matches_reference_mcts_py(case):
    logic = orc.GameLogic(abi.rules.BY_NAME[case["rules"]], case["side_len"])
    st = state_from_case(case)
    assert st.to_fen() == case["fen"]
    kids, root_ns, stats = logic.mcts(st, case["n_sims"], case["cpuct"], case["seed"], case["max_plies"],
                                      game_id=case["game_id"])
    assert root_ns == case["root_ns"]
    got = [[a, n, float(q).hex()] for (_p, a, n, q) in kids]
    assert got == case["root_children"]
    assert stats.sims == case["n_sims"]
    # visited root children are a prefix of the canonical legal list (DESIGN.md "tree layout")
    legal = [abi.action_encode(case["side_len"], p) for p in logic.all_plays(st)]
    assert [g[0] for g in got] == legal[:len(got)]
    # probs of mcts.py:48-53 (temp = 1)
    total = float(sum(n for _, n, _ in got))
    probs = [(a, (n / total).hex()) for a, n, _ in got if n]
    assert probs == [tuple(x) for x in case["probs_temp1_nonzero"]]


def test_config1_plumbing_reports_rate():
    """BASELINE.json configs[0]: single 7x7 Brandubh game, 1000-sim random-rollout MCTS on the CPU path."""
    import time
    case = next(c for c in GOLD["cases"] if c["name"] == "config1_brandubh_1000")
    logic = orc.GameLogic(abi.rules.BRANDUBH, 7)
    st = state_from_case(case)
    t0 = time.perf_counter()
    kids, root_ns, stats = logic.mcts(st, 1000, 1.0, 0, 256)
    dt = time.perf_counter() - t0
    assert root_ns == 999 and stats.rollouts == 1000
    print(f"config1: {1000 / dt:.0f} sims/s, {stats.rollout_plies / dt:.0f} rollout plies/s on 1 CPU thread")
