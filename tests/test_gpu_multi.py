"""GPU tests of what N > 1 relies on (BASELINE.json configs[3]): independent contexts, streams and
game-id shards.  Games never interact (game/game/logic.rs:61-65), so a shard's results must equal the oracle's for the same
GLOBAL ids whatever else runs beside it.  Needs a real MI355X: `pytest -m gpu`."""
import ctypes as C
import threading

import pytest

from alphazeroforhnefatafl_amd import abi
from alphazeroforhnefatafl_amd.abi import TaflMctsParams
from oracle import oracle as orc
from tests import parity_util as pu

pytestmark = pytest.mark.gpu


def _children(kids, cnt, g, width):
    return [(kids[g * width + j].action, kids[g * width + j].visits, float(kids[g * width + j].q).hex()) for j in range(cnt[g])]


def _oracle_children(lg, fen, rules, wb, p, gid):
    one = pu.start_states(orc, fen, rules.starting_side, wb, 1)
    ok, on, _ = orc.batch_mcts(lg, one, 1, wb, p, gid)
    return [(ok[j].action, ok[j].visits, float(ok[j].q).hex()) for j in range(on[0])]


def test_two_contexts_from_two_threads_run_concurrently():
    """INTEGRATION.md section 4 (handles are thread-compatible, contexts independent): two tafl_ctx with their own streams on
    device 0, two 65 536-game batches with game_id_base 0 and 7 * 65 536, searched at the same time from two host threads.
    Each must equal (i) the oracle on scattered global ids and (ii) the same shard searched alone."""
    from alphazeroforhnefatafl_amd.engine import BatchedGameLogic
    rules, fen, wb = pu.CONFIGS["copenhagen11"]
    n, G, sims, cap, seed, W = 11, 65536, 12, 192, 2, 16
    bases = [0, 7 * 65536]
    logics = [BatchedGameLogic(rules, n, wb, device=0) for _ in bases]       # stream=None: each ctx creates its own stream
    assert logics[0]._h.value != logics[1]._h.value
    batches = [lg.new_batch(G, fen) for lg in logics]
    for b in batches:
        b.mcts_reserve(sims)
    errs, out = [], [None, None]

    def worker(i):
        try:
            for _ in range(3):                                               # several back-to-back searches keep both streams busy together
                batches[i].mcts_run(sims, 1.0, seed, cap, game_id_base=bases[i])
            out[i] = (batches[i].mcts_root_children(W), batches[i].mcts_stats())
        except Exception as e:   # noqa: BLE001
            errs.append((i, repr(e)))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    olg = orc.GameLogic(rules, n)
    p = TaflMctsParams(sims, cap, 1.0, seed, 0, 0)
    for i, base in enumerate(bases):
        (kids, cnt), st = out[i]
        assert st.sims == G * sims and st.faults == 0
        for g in (0, 1, 63, 64, 31337, 65535):
            assert _children(kids, cnt, g, W) == _oracle_children(olg, fen, rules, wb, p, base + g), (base, g)
    # the same shards searched alone, one after the other, on a third context
    solo = BatchedGameLogic(rules, n, wb, device=0)
    sb = solo.new_batch(G, fen)
    for i, base in enumerate(bases):
        sb.mcts_run(sims, 1.0, seed, cap, game_id_base=base)
        kids, cnt = sb.mcts_root_children(W)
        (ck, cc), _ = out[i]
        assert list(cnt) == list(cc)
        assert bytes(kids) == bytes(ck), base
    assert bytes(out[0][0][0]) != bytes(out[1][0][0])                        # different ids, different searches
    for b in batches + [sb]:
        b.close()
    for lg in logics + [solo]:
        lg.close()


def test_config3_last_shard_id_range_at_bench_settings():
    """BASELINE.json configs[3] = 524 288 games over 8 GPUs: rank 7's shard (global ids 458 752 .. 524 287) at the bench's own
    settings (65 536 games, S = 64, cap 512, seed 2) against the oracle on scattered ids, plus the step's conservation laws."""
    from alphazeroforhnefatafl_amd.engine import BatchedGameLogic
    rules, fen, wb = pu.CONFIGS["copenhagen11"]
    n, G, sims, cap, seed, W = 11, 65536, 64, 512, 2, 72
    base = 7 * 65536
    lg = BatchedGameLogic(rules, n, wb, device=0)
    b = lg.new_batch(G, fen)
    b.mcts_run(sims, 1.0, seed, cap, game_id_base=base)
    st = b.mcts_stats()
    assert st.sims == G * sims and st.faults == 0
    assert st.rollouts + st.terminal_hits == st.sims and sum(st.reason_hist) == st.rollouts
    kids, cnt = b.mcts_root_children(W)
    for g in range(0, G, 127):
        assert sum(kids[g * W + j].visits for j in range(cnt[g])) == sims - 1
    olg = orc.GameLogic(rules, n)
    p = TaflMctsParams(sims, cap, 1.0, seed, 0, 0)
    for g in (0, 4095, 40000, 65535):
        assert _children(kids, cnt, g, W) == _oracle_children(olg, fen, rules, wb, p, base + g), g
    # ids beyond 32 bits reach the RNG key too (game_id is 64-bit in the ABI)
    big = (1 << 40) + 12345
    small = lg.new_batch(64, fen)
    small.mcts_run(16, 1.0, seed, 128, game_id_base=big)
    sk, sc = small.mcts_root_children(W)
    p2 = TaflMctsParams(16, 128, 1.0, seed, 0, 0)
    for g in (0, 63):
        assert _children(sk, sc, g, W) == _oracle_children(olg, fen, rules, wb, p2, big + g), g
    small.close(); b.close(); lg.close()
