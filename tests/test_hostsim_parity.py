"""Differential tests: the product's bit-parallel engine (device code compiled for the host,
tests/hostsim) against the literal oracle, on seeded random games and synthetic positions.
CPU only.  The same comparisons run against the real HIP kernels in tests/test_gpu_parity.py.
"""
import ctypes as C
import random

import pytest

from alphazeroforhnefatafl_amd import abi
from alphazeroforhnefatafl_amd.abi import TaflMctsParams
from oracle import oracle as orc
from tests.hostsim.hostsim import HostSim
from tests import parity_util as pu


def _mk(name):
    rules, fen, wb = pu.CONFIGS[name]
    n = abi.fen_side_len(fen)
    return rules, fen, wb, n, orc.GameLogic(rules, n), HostSim(rules, n, wb)


@pytest.mark.parametrize("name", list(pu.CONFIGS))
def test_lockstep_games_step_kth(name):
    """Play seeded random games ply by ply; at every ply compare legal-move masks and the full post-state."""
    rules, fen, wb, n, lg, hs = _mk(name)
    G, T = 96, 220
    rng = random.Random(1234)
    states = pu.start_states(orc, fen, rules.starting_side, wb, G)
    for t in range(T):
        oc, om = orc.batch_movegen(lg, states, G, wb)
        hc, hm = hs.movegen(states, G)
        assert list(oc) == list(hc), (name, t)
        assert bytes(om) == bytes(hm), (name, t)
        ranks = (C.c_uint32 * G)(*[rng.randrange(1 << 30) for _ in range(G)])
        a = pu.clone_states(states, G)
        b = pu.clone_states(states, G)
        op, oe = orc.batch_step_kth(lg, a, G, wb, ranks)
        hp, he = hs.step_kth(b, G, ranks)
        for g in range(G):
            assert pu.play_tuple4(op[g]) == pu.play_tuple4(hp[g]), (name, t, g, pu.describe_state(states[g], wb))
            assert pu.effects_tuple(oe[g]) == pu.effects_tuple(he[g]), (name, t, g, pu.describe_state(states[g], wb), pu.play_tuple4(op[g]))
        if not pu.states_equal(a, b, G):
            g = pu.first_state_diff(a, b, G)
            raise AssertionError((name, t, g, pu.describe_state(states[g], wb), pu.play_tuple4(op[g]),
                                  pu.describe_state(a[g], wb), pu.describe_state(b[g], wb)))
        states = a
        # restart finished games so the batch keeps exercising play
        if t % 50 == 49:
            fresh = pu.start_states(orc, fen, rules.starting_side, wb, 1)[0]
            for g in range(G):
                if states[g].status != abi.ONGOING:
                    C.memmove(C.byref(states, g * C.sizeof(abi.TaflState)), C.byref(fresh), C.sizeof(abi.TaflState))


@pytest.mark.parametrize("name", ["copenhagen11", "brandubh7", "tablut9", "magpie7", "copenhagen13", "koch7"])
def test_synthetic_positions(name):
    """Random (unreachable) positions: masks, validate codes for arbitrary plays, one step, side_can_play."""
    rules, fen, wb, n, lg, hs = _mk(name)
    rng = random.Random(99)
    G = 1500
    states = pu.random_board_states(rng, n, wb, G)
    oc, om = orc.batch_movegen(lg, states, G, wb)
    hc, hm = hs.movegen(states, G)
    for g in range(G):
        assert oc[g] == hc[g], (name, g, pu.describe_state(states[g], wb))
    assert bytes(om) == bytes(hm)
    plays = pu.random_plays(rng, n, G)
    codes = hs.validate(states, G, plays)
    for g in range(G):
        st = orc.GameState.from_abi(states[g], wb)
        assert lg.validate_play(plays[g], st) == codes[g], (name, g, pu.describe_state(states[g], wb), pu.play_tuple4(plays[g]))
    for side in (abi.ATTACKER, abi.DEFENDER):
        out = hs.side_can_play(states, G, side)
        for g in range(0, G, 7):
            st = orc.GameState.from_abi(states[g], wb)
            assert lg.side_can_play(side, st) == bool(out[g]), (name, g, side)
    # arbitrary (mostly invalid) plays through do_play
    a = pu.clone_states(states, G)
    b = pu.clone_states(states, G)
    oe = orc.batch_step(lg, a, G, wb, plays)
    he = hs.step(b, G, plays)
    for g in range(G):
        assert pu.effects_tuple(oe[g]) == pu.effects_tuple(he[g]), (name, g)
    assert pu.states_equal(a, b, G)
    # three legal plies from each synthetic position
    for t in range(3):
        ranks = (C.c_uint32 * G)(*[rng.randrange(1 << 30) for _ in range(G)])
        a = pu.clone_states(states, G)
        b = pu.clone_states(states, G)
        op, oe = orc.batch_step_kth(lg, a, G, wb, ranks)
        hp, he = hs.step_kth(b, G, ranks)
        for g in range(G):
            assert pu.play_tuple4(op[g]) == pu.play_tuple4(hp[g]), (name, t, g, pu.describe_state(states[g], wb))
            assert pu.effects_tuple(oe[g]) == pu.effects_tuple(he[g]), (name, t, g, pu.describe_state(states[g], wb), pu.play_tuple4(op[g]))
        if not pu.states_equal(a, b, G):
            g = pu.first_state_diff(a, b, G)
            raise AssertionError((name, t, g, pu.describe_state(states[g], wb), pu.play_tuple4(op[g]),
                                  pu.describe_state(a[g], wb), pu.describe_state(b[g], wb)))
        states = a


@pytest.mark.parametrize("name", ["copenhagen11", "brandubh7", "tablut9", "copenhagen13"])
def test_rollouts(name):
    rules, fen, wb, n, lg, hs = _mk(name)
    G = 200
    states = pu.start_states(orc, fen, rules.starting_side, wb, G)
    plies = (C.c_uint32 * G)(*[i % 64 for i in range(G)])
    a = pu.clone_states(states, G)
    b = pu.clone_states(states, G)
    orc.batch_random_advance(lg, a, G, wb, 1, plies, 1000)
    hs.random_advance(b, G, 1, plies, 1000)
    assert pu.states_equal(a, b, G)
    ro = orc.batch_rollout(lg, a, G, wb, 5, 3, 300, 1000)
    rh = hs.rollout(a, G, 5, 3, 300, 1000)
    for g in range(G):
        assert (ro[g].value, ro[g].status, ro[g].reason, ro[g].winner, ro[g].plies) == \
               (rh[g].value, rh[g].status, rh[g].reason, rh[g].winner, rh[g].plies), (name, g)


@pytest.mark.parametrize("name,sims,cpuct", [("copenhagen11", 48, 1.0), ("brandubh7", 200, 1.0), ("tablut9", 64, 1.5),
                                             ("copenhagen13", 24, 1.0),
                                             # c_puct == 0: an unvisited action scores 0 and TIES a visited one with Qsa == 0, which then wins
                                             # by its lower index (mcts.py:117-119): puct_pick's Qsa > 0 shortcut must not be taken
                                             ("brandubh7", 300, 0.0), ("copenhagen11", 200, 0.0), ("brandubh7", 120, 1e-300)])
def test_mcts(name, sims, cpuct):
    rules, fen, wb, n, lg, hs = _mk(name)
    G = 24
    states = pu.start_states(orc, fen, rules.starting_side, wb, G)
    plies = (C.c_uint32 * G)(*[(i * 5) % 40 for i in range(G)])
    orc.batch_random_advance(lg, states, G, wb, 11, plies, 77)
    p = TaflMctsParams(sims, 200, cpuct, 42, 0, 0)
    ok, on, ostats = orc.batch_mcts(lg, states, G, wb, p, 77)
    hk, hn, hstats = hs.mcts(states, G, p, 77)
    assert list(on) == list(hn)
    for g in range(G):
        for j in range(on[g]):
            a, b = ok[g * 256 + j], hk[g * 256 + j]
            assert (pu.play_tuple4(a.play), a.action, a.visits, float(a.q).hex()) == \
                   (pu.play_tuple4(b.play), b.action, b.visits, float(b.q).hex()), (name, g, j)
    for f in ("sims", "rollouts", "rollout_plies", "tree_depth_sum", "children_scanned", "terminal_hits", "faults"):
        assert getattr(ostats, f) == getattr(hstats, f), f
    assert list(ostats.reason_hist) == list(hstats.reason_hist)


@pytest.mark.parametrize("name,sims,k,target", [("copenhagen13", 190, 8, 4), ("copenhagen13", 170, 4, 0), ("copenhagen11", 300, 8, 4), ("brandubh7", 500, 8, 2)])
def test_mcts_wide_roots_and_deep_trees(name, sims, k, target):
    """The tree step keeps the root header and the sign bits of the first 128 root edges in registers (tafl_ops.hpp RootCache): searches
    whose root grows past 128 children (13x13: 152 legal plays at the start), that visit every root child and go on into deep trees, and
    that run long enough for every edge array to move several times - against the oracle, bit for bit."""
    from tests.hostsim import hostsim
    rules, fen, wb, n, lg, hs = _mk(name)
    G = 3
    states = pu.start_states(orc, fen, rules.starting_side, wb, G)
    hostsim.set_spec_k(k, target, 0)
    try:
        p = TaflMctsParams(sims, 48, 1.0, 7, 0, 0)
        ok, on, ostats = orc.batch_mcts(lg, states, G, wb, p, 0)
        hk, hn, hstats = hs.mcts(states, G, p, 0)
    finally:
        hostsim.set_spec_k(4, 0, 0)
    assert list(on) == list(hn)
    if name == "copenhagen13":
        assert max(on) > 128
    for g in range(G):
        for j in range(on[g]):
            a, b = ok[g * 256 + j], hk[g * 256 + j]
            assert (a.action, a.visits, float(a.q).hex()) == (b.action, b.visits, float(b.q).hex()), (name, g, j)
    for f in ("sims", "rollouts", "rollout_plies", "tree_depth_sum", "children_scanned", "terminal_hits", "faults"):
        assert getattr(ostats, f) == getattr(hstats, f), f


@pytest.mark.parametrize("k,cooldown", [(1, 0), (2, 0), (2, 1), (3, 2), (4, 0), (4, 4), (8, 0), (8, 3), (8, 8)])
def test_mcts_speculative_slots_do_not_change_results(k, cooldown):
    """The MCTS pipeline with k playout slots per game (k-1 predicted simulations, `cooldown` = slots per round the search is planned
    for) must reproduce the sequential search exactly: root statistics as float64 bit patterns and every counter, for start,
    mid-game and terminal-heavy positions.  The host-sim driver also checks that the speculation pass leaves no trace in the tree."""
    import json
    import os
    from tests.hostsim import hostsim
    hostsim.set_spec_k(k, cooldown)
    try:
        gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mcts_golden.json")))
        for case in gold["cases"]:
            if case["n_sims"] > 300:
                continue
            rules = abi.rules.BY_NAME[case["rules"]]
            st = abi.TaflState.from_buffer_copy(bytes.fromhex(case["state_hex"]))
            hs = HostSim(rules, case["side_len"], case["word_bits"])
            arr = (abi.TaflState * 2)(st, st)
            p = TaflMctsParams(case["n_sims"], case["max_plies"], case["cpuct"], case["seed"], 0, 0)
            kids, cnt, stats = hs.mcts(arr, 2, p, case["game_id"])
            got = [[kids[j].action, kids[j].visits, float(kids[j].q).hex()] for j in range(cnt[0])]
            assert got == case["root_children"], (k, case["name"])
            assert stats.faults == 0 and stats.sims == 2 * case["n_sims"]
            if k == 1:
                assert stats.spec_issued == 0 and stats.spec_hits == 0
        # terminal-heavy synthetic positions against the oracle, incl. all counters
        rng = random.Random(5)
        n, wb = 7, 64
        lst = pu.sparse_endgame_positions(rng, n, wb, 40)
        states = pu.states_array(lst)
        lg, hs = orc.GameLogic(abi.rules.BRANDUBH, n), HostSim(abi.rules.BRANDUBH, n, wb)
        p = TaflMctsParams(60, 40, 1.0, 3, 0, 0)
        ok, on, ostats = orc.batch_mcts(lg, states, 40, wb, p, 5)
        hk, hn, hstats = hs.mcts(states, 40, p, 5)
        assert list(on) == list(hn)
        for g in range(40):
            for j in range(on[g]):
                a, b = ok[g * 256 + j], hk[g * 256 + j]
                assert (a.action, a.visits, float(a.q).hex()) == (b.action, b.visits, float(b.q).hex()), (k, g, j)
        for f in ("sims", "rollouts", "rollout_plies", "tree_depth_sum", "children_scanned", "terminal_hits", "faults"):
            assert getattr(ostats, f) == getattr(hstats, f), (k, f)
        assert list(ostats.reason_hist) == list(hstats.reason_hist)
        assert ostats.terminal_hits > 0
    finally:
        hostsim.set_spec_k(4, 0, 0)


@pytest.mark.parametrize("k,target,capacity", [(8, 4, 70), (8, 0, 40), (4, 4, 33), (8, 8, 1), (2, 2, 20)])
def test_mcts_with_a_full_device_round_capacity(k, target, capacity):
    """More playouts requested than one round may run (the device holds occupancy x CUs x 64 lanes at once): the rest waits for the
    next round, the pending leaf of every waiting game first.  Same results as the sequential search, bit for bit."""
    from tests.hostsim import hostsim
    rules, fen, wb = pu.CONFIGS["brandubh7"]
    n, G = 7, 32
    lg, hs = orc.GameLogic(rules, n), HostSim(rules, n, wb)
    states = pu.start_states(orc, fen, rules.starting_side, wb, G)
    plies = (C.c_uint32 * G)(*[(i * 7) % 30 for i in range(G)])
    orc.batch_random_advance(lg, states, G, wb, 23, plies, 9)
    p = TaflMctsParams(48, 64, 1.25, 6, 0, 0)
    ok, on, ostats = orc.batch_mcts(lg, states, G, wb, p, 9)
    hostsim.set_spec_k(k, target, capacity)
    try:
        hk, hn, hstats = hs.mcts(states, G, p, 9)
        work = hostsim.round_work()
    finally:
        hostsim.set_spec_k(4, 0, 0)
    assert max(work) <= capacity
    assert list(on) == list(hn)
    for g in range(G):
        for j in range(on[g]):
            a, b = ok[g * 256 + j], hk[g * 256 + j]
            assert (a.action, a.visits, float(a.q).hex()) == (b.action, b.visits, float(b.q).hex()), (g, j)
    for f in ("sims", "rollouts", "rollout_plies", "tree_depth_sum", "children_scanned", "terminal_hits", "faults"):
        assert getattr(ostats, f) == getattr(hstats, f), f
    assert list(ostats.reason_hist) == list(hstats.reason_hist)


@pytest.mark.parametrize("scen", [1, 2])
@pytest.mark.parametrize("name,sims,capacity", [("brandubh7", 90, 0), ("copenhagen11", 70, 0), ("brandubh7", 60, 50), ("copenhagen13", 24, 0)])
def test_mcts_prediction_scenarios_do_not_change_results(name, sims, capacity, scen):
    """The prediction pass runs one or two scenarios for the pending playout's value (no decision / the side that has won more of
    the search's playouts wins): which, in which order and how many is a policy (Ops::mcts_scenarios).  Whatever it is, the search
    equals the sequential one bit for bit - positions at several phases of a game, with and without a full device."""
    from tests.hostsim import hostsim
    rules, fen, wb = pu.CONFIGS[name]
    n, G = abi.fen_side_len(fen), 12
    lg, hs = orc.GameLogic(rules, n), HostSim(rules, n, wb)
    states = pu.start_states(orc, fen, rules.starting_side, wb, G)
    plies = (C.c_uint32 * G)(*[(i * 11) % 60 for i in range(G)])
    orc.batch_random_advance(lg, states, G, wb, 31, plies, 4)
    p = TaflMctsParams(sims, 96, 1.0, 8, 0, 0)
    ok, on, ostats = orc.batch_mcts(lg, states, G, wb, p, 4)
    hostsim.set_spec_k(8, 4, capacity)
    hostsim.set_scenarios(scen)
    try:
        hk, hn, hstats = hs.mcts(states, G, p, 4)
    finally:
        hostsim.set_spec_k(4, 0, 0)
        hostsim.set_scenarios(0)
    assert list(on) == list(hn)
    for g in range(G):
        for j in range(on[g]):
            a, b = ok[g * 256 + j], hk[g * 256 + j]
            assert (a.action, a.visits, float(a.q).hex()) == (b.action, b.visits, float(b.q).hex()), (g, j)
    for f in ("sims", "rollouts", "rollout_plies", "tree_depth_sum", "children_scanned", "terminal_hits", "faults"):
        assert getattr(ostats, f) == getattr(hstats, f), f
    assert list(ostats.reason_hist) == list(hstats.reason_hist)


@pytest.mark.parametrize("name,sims,k", [("brandubh7", 150, 4), ("copenhagen11", 140, 8), ("tablut9", 120, 1)])
def test_mcts_first_play_urgency_flag(name, sims, k):
    """TAFL_MCTS_FLAG_FPU_INF (the src/mcts.rs sketch: unvisited actions score +inf, mcts.rs:49-51; a new node starts with visits 1,
    mcts.rs:187): the device functions against the oracle's twin, bit for bit.  The Rust sketch cannot be built or run, so this mode is
    pinned by the oracle alone.  Every root child is expanded before any is revisited."""
    from tests.hostsim import hostsim
    rules, fen, wb = pu.CONFIGS[name]
    n = abi.fen_side_len(fen)
    G = 6
    lg, hs = orc.GameLogic(rules, n), HostSim(rules, n, wb)
    states = pu.start_states(orc, fen, rules.starting_side, wb, G)
    plies = (C.c_uint32 * G)(*[(i * 4) % 13 for i in range(G)])
    orc.batch_random_advance(lg, states, G, wb, 3, plies, 2)
    p = TaflMctsParams(sims, 200, 1.0, 8, 0, abi.MCTS_FLAG_FPU_INF)
    p0 = TaflMctsParams(sims, 200, 1.0, 8, 0, 0)
    ok, on, ostats = orc.batch_mcts(lg, states, G, wb, p, 2)
    ok0, on0, _ = orc.batch_mcts(lg, states, G, wb, p0, 2)
    hostsim.set_spec_k(k, 0, 0)
    try:
        hk, hn, hstats = hs.mcts(states, G, p, 2)
    finally:
        hostsim.set_spec_k(4, 0, 0)
    assert list(on) == list(hn)
    counts, _ = orc.batch_movegen(lg, states, G, wb, want_masks=False)
    differs = False
    for g in range(G):
        if sims > counts[g] and counts[g] > 0:
            assert on[g] == counts[g], (g, on[g], counts[g])          # breadth first: every legal play of the root was tried
        for j in range(on[g]):
            a, b = ok[g * 256 + j], hk[g * 256 + j]
            assert (a.action, a.visits, float(a.q).hex()) == (b.action, b.visits, float(b.q).hex()), (g, j)
        differs |= [(ok[g * 256 + j].action, ok[g * 256 + j].visits) for j in range(on[g])] != [(ok0[g * 256 + j].action, ok0[g * 256 + j].visits) for j in range(on0[g])]
    assert differs
    for f in ("sims", "rollouts", "rollout_plies", "tree_depth_sum", "children_scanned", "terminal_hits", "faults"):
        assert getattr(ostats, f) == getattr(hstats, f), f


@pytest.mark.parametrize("log_cap", [1, 2, 5, 16])
def test_mcts_undo_log_overflow_only_ends_the_prediction_pass(log_cap):
    """The prediction pass keeps its undo log in a small per-lane scratch (LDS on the device: 16 records of each kind, 5 in the fused
    kernel).  A pass that runs out of records stops predicting and restores what it changed: fewer predictions, identical results."""
    from tests.hostsim import hostsim
    hostsim.set_spec_k(8, 4, 0)
    hostsim.set_log_cap(log_cap)
    try:
        for name, G, sims, plies_cap in (("brandubh7", 12, 220, 80), ("copenhagen11", 6, 90, 60)):
            rules, fen, wb, n, lg, hs = _mk(name)
            states = pu.start_states(orc, fen, rules.starting_side, wb, G)
            plies = (C.c_uint32 * G)(*[(i * 7) % 30 for i in range(G)])
            orc.batch_random_advance(lg, states, G, wb, 11, plies, 5)
            p = TaflMctsParams(sims, plies_cap, 1.0, 17, 0, 0)
            ok, on, ostats = orc.batch_mcts(lg, states, G, wb, p, 5)
            hk, hn, hstats = hs.mcts(states, G, p, 5)
            assert list(on) == list(hn)
            for g in range(G):
                for j in range(on[g]):
                    a, b = ok[g * 256 + j], hk[g * 256 + j]
                    assert (a.action, a.visits, float(a.q).hex()) == (b.action, b.visits, float(b.q).hex()), (name, log_cap, g, j)
            for f in ("sims", "rollouts", "rollout_plies", "tree_depth_sum", "children_scanned", "terminal_hits", "faults"):
                assert getattr(ostats, f) == getattr(hstats, f), (name, f)
    finally:
        hostsim.set_spec_k(4, 0, 0)
        hostsim.set_log_cap(16)


def _oracle_selfplay(lg, states, G, wb, sims, cap, cpuct, seed, base, n_moves):
    """The loop tafl_selfplay_run replaces, on the oracle: per move a search with sim_offset = move * sims, then the most visited root
    play (first maximum, src/mcts.rs:216-227) on every game that has one."""
    plays_all = []
    for m in range(n_moves):
        p = TaflMctsParams(sims, cap, cpuct, seed, m * sims, 0)
        kids, cnt, _ = orc.batch_mcts(lg, states, G, wb, p, base)
        sub = (abi.TaflPlay * G)()
        row = []
        for g in range(G):
            vs = [kids[g * 256 + j].visits for j in range(cnt[g])]
            if vs and max(vs) > 0 and states[g].status == 0:
                best = kids[g * 256 + vs.index(max(vs))].play
                C.memmove(C.byref(sub[g]), C.byref(best), C.sizeof(abi.TaflPlay))
                row.append(pu.play_tuple4(best))
            else:
                row.append((0, 0, 0, 0))
        orc.batch_step(lg, states, G, wb, sub)           # (an all-zero play on a finished game is rejected and changes nothing)
        plays_all.append(row)
    return plays_all


@pytest.mark.parametrize("name,G,sims,n_moves,k,target,cap", [("brandubh7", 24, 40, 9, 8, 4, 0), ("copenhagen11", 8, 24, 4, 8, 4, 20), ("tablut9", 10, 30, 5, 4, 2, 0),
                                                             ("copenhagen13", 4, 16, 3, 8, 4, 0)])
def test_selfplay_run_equals_the_loop_of_searches_and_plays(name, G, sims, n_moves, k, target, cap):
    """tafl_selfplay_run's per-game state machine (Ops::selfplay_advance: most visited play on the batch state, next search at once, RNG key
    of move m offset by m * n_sims, plan counted from the launch the game's own search began) == the synchronous loop on the oracle: plays
    and final states, for games that end on the way too (Brandubh playouts are short and decisive: several of the 24 games finish)."""
    from tests.hostsim import hostsim
    rules, fen, wb, n, lg, hs = _mk(name)
    states = pu.start_states(orc, fen, rules.starting_side, wb, G)
    plies = (C.c_uint32 * G)(*[(i * 7) % 36 for i in range(G)])
    orc.batch_random_advance(lg, states, G, wb, 3, plies, 50)
    want_states = pu.clone_states(states, G)
    want = _oracle_selfplay(lg, want_states, G, wb, sims, 60, 1.0, 8, 50, n_moves)
    hostsim.set_spec_k(k, target, cap)
    if name == "copenhagen13":
        hostsim.set_dense13(True)
    try:
        got_states = pu.clone_states(states, G)
        plays, stats = hs.selfplay(got_states, G, TaflMctsParams(sims, 60, 1.0, 8, 0, 0), n_moves, 50)
    finally:
        hostsim.set_spec_k(4, 0, 0)
        hostsim.set_dense13(False)
    for m in range(n_moves):
        assert [pu.play_tuple4(plays[m * G + g]) for g in range(G)] == want[m], (name, m)
    assert pu.states_equal(want_states, got_states, G), pu.first_state_diff(want_states, got_states, G)
    assert stats.faults == 0
