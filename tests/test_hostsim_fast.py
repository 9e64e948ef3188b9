"""The fast playout engine (csrc/tafl_fast.hpp: two-layout subtraction move generator + enclosure filter) against the
oracle AND against the generic engine, from reachable, synthetic and crafted positions and under random rulesets.
Playouts from crafted positions end quickly in enclosure / exit-fort / shieldwall / no-plays outcomes, so the filter
and every rare rule are exercised.  CPU only (device code compiled for the host)."""
import collections
import ctypes as C
import random

import pytest

from alphazeroforhnefatafl_amd import abi
from oracle import oracle as orc
from tests.hostsim import hostsim
from tests.hostsim.hostsim import HostSim
from tests import parity_util as pu


def _tuple(r):
    return (r.value, r.status, r.reason, r.winner, r.plies)


def _compare(rules, n, wb, states, G, seed, cap, tag):
    lg, hs = orc.GameLogic(rules, n), HostSim(rules, n, wb)
    ro = orc.batch_rollout(lg, states, G, wb, seed, 3, cap, 100)
    hostsim.force_generic(False)
    rf = hs.rollout(states, G, seed, 3, cap, 100)
    hostsim.force_generic(True)
    rg = hs.rollout(states, G, seed, 3, cap, 100)
    hostsim.force_generic(False)
    hist = collections.Counter()
    for g in range(G):
        assert _tuple(ro[g]) == _tuple(rf[g]), (tag, "fast", g, pu.describe_state(states[g], wb), _tuple(ro[g]), _tuple(rf[g]))
        assert _tuple(ro[g]) == _tuple(rg[g]), (tag, "generic", g)
        hist[ro[g].reason] += 1
    return hist


@pytest.mark.parametrize("name", ["copenhagen11", "brandubh7", "tablut9", "copenhagen13", "koch7", "copenhagen9_u256", "magpie7"])
def test_fast_rollouts_from_reachable_positions(name):
    rules, fen, wb = pu.CONFIGS[name]
    n = abi.fen_side_len(fen)
    lg = orc.GameLogic(rules, n)
    G = 160
    states = pu.start_states(orc, fen, rules.starting_side, wb, G)
    plies = (C.c_uint32 * G)(*[(i * 3) % 70 for i in range(G)])
    orc.batch_random_advance(lg, states, G, wb, 21, plies, 9)
    _compare(rules, n, wb, states, G, 8, 400, name)
    # random_advance itself (in-place fast playout)
    a = pu.start_states(orc, fen, rules.starting_side, wb, G)
    b = pu.clone_states(a, G)
    orc.batch_random_advance(lg, a, G, wb, 5, plies, 0)
    HostSim(rules, n, wb).random_advance(b, G, 5, plies, 0)
    assert pu.states_equal(a, b, G), name


def test_fast_rollouts_cover_rare_outcomes():
    rng = random.Random(31)
    total = collections.Counter()
    for rname, n, wb in [("copenhagen", 11, 128), ("copenhagen", 9, 128), ("copenhagen", 13, 256), ("brandubh", 7, 64), ("koch", 7, 128), ("tablut", 9, 128)]:
        rules = abi.rules.BY_NAME[rname]
        lst = pu.enclosure_positions(rng, n, wb, 120) + pu.shieldwall_positions(rng, n, wb, 120) + \
            pu.sparse_endgame_positions(rng, n, wb, 120) + [s for s in pu.random_board_states(rng, n, wb, 120)]
        if n >= 9:
            lst += pu.exit_fort_positions(rng, n, wb, 120)
        states = pu.states_array(lst)
        for seed in (1, 2):
            total += _compare(rules, n, wb, states, len(lst), seed, 80, (rname, n))
    assert total[abi.ENCLOSED] >= 10, total
    assert total[abi.EXIT_FORT] >= 10, total
    assert total[abi.KING_CAPTURED] >= 10 and total[abi.KING_ESCAPED] >= 10, total
    assert total[abi.WIN_NO_PLAYS] + total[8 + abi.DRAW_NO_PLAYS] >= 3, total


@pytest.mark.parametrize("seed", range(16))
def test_fast_rollouts_random_rulesets(seed):
    rng = random.Random(5000 + seed)
    rules = pu.random_ruleset(rng)
    n, wb = rng.choice([(7, 64), (9, 128), (11, 128), (13, 256)])
    lst = [s for s in pu.random_board_states(rng, n, wb, 80)] + pu.enclosure_positions(rng, n, wb, 40) + \
        pu.shieldwall_positions(rng, n, wb, 40) + pu.sparse_endgame_positions(rng, n, wb, 40)
    if n >= 9:
        lst += pu.exit_fort_positions(rng, n, wb, 40)
    _compare(rules, n, wb, pu.states_array(lst), len(lst), seed, 60, ("fuzz", seed))


@pytest.mark.parametrize("n,wb", [(11, 128), (9, 128), (13, 256), (7, 64)])
def test_fast_shieldwall_hint_many_seeds(n, wb):
    """Very short playouts from wall-ready positions under many seeds: the wall-completing play is drawn often, so a
    false negative of the fast engine's window pre-filter would show up as a different capture count / outcome."""
    rng = random.Random(77)
    rules = abi.rules.COPENHAGEN
    variants = (rules, rules.replace(shieldwall=(False, abi.ps_all())))
    lst = pu.shieldwall_positions(rng, n, wb, 150)
    states = pu.states_array(lst)
    lg0 = orc.GameLogic(rules, n)
    # make sure the workload really contains wall captures: count multi-captures among all legal plays (oracle)
    oc, _ = orc.batch_movegen(lg0, states, len(lst), wb)
    arr, ranks, total, _ = pu.expand_all(states, len(lst), oc)
    _, oe = orc.batch_step_kth(lg0, pu.clone_states(arr, total), total, wb, ranks)
    assert sum(1 for i in range(total) if oe[i].n_captures >= 2) >= 20
    for v in variants:
        for seed in range(12):
            _compare(v, n, wb, states, len(lst), 100 + seed, 3, ("swhint", n, seed))


def test_dense_13_column_layout_matches_the_oracle():
    """13x13 positions searched in the dense 13-column layout (6 limbs) that the library uses for the 13x13 preset: rollouts and MCTS
    root statistics equal the oracle's (which works on the reference's U256 / 15-column words), from the start and from mid-game."""
    from alphazeroforhnefatafl_amd.abi import TaflMctsParams
    from tests.hostsim import hostsim
    rules, fen, wb = pu.CONFIGS["copenhagen13"]
    n, G = 13, 24
    lg, hs = orc.GameLogic(rules, n), HostSim(rules, n, wb)
    states = pu.start_states(orc, fen, rules.starting_side, wb, G)
    plies = (C.c_uint32 * G)(*[(i * 11) % 90 for i in range(G)])
    orc.batch_random_advance(lg, states, G, wb, 5, plies, 3)
    want = orc.batch_rollout(lg, states, G, wb, 9, 2, 400, 3)
    p = TaflMctsParams(40, 300, 1.0, 4, 0, 0)
    ok, on, ostats = orc.batch_mcts(lg, states, G, wb, p, 3)
    hostsim.set_dense13(True)
    try:
        got = hs.rollout(states, G, 9, 2, 400, 3)
        hk, hn, hstats = hs.mcts(states, G, p, 3)
    finally:
        hostsim.set_dense13(False)
    for g in range(G):
        assert (want[g].value, want[g].status, want[g].reason, want[g].winner, want[g].plies) == \
               (got[g].value, got[g].status, got[g].reason, got[g].winner, got[g].plies), g
    assert list(on) == list(hn)
    for g in range(G):
        for j in range(on[g]):
            a, b = ok[g * 256 + j], hk[g * 256 + j]
            assert (pu.play_tuple4(a.play), a.action, a.visits, float(a.q).hex()) == (pu.play_tuple4(b.play), b.action, b.visits, float(b.q).hex()), (g, j)
    for f in ("sims", "rollouts", "rollout_plies", "tree_depth_sum", "children_scanned", "terminal_hits", "faults"):
        assert getattr(ostats, f) == getattr(hstats, f), f
