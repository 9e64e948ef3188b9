"""Replay-buffer text format: libtaflhip's host-side writer vs the statement-by-statement restatement
of write_to_file (game/main.rs:86-132) in oracle/oracle.py — byte-identical files, including the reference's line-wise FIFO.
The reference never calls write_to_file and holds no fixture for it: parity is pinned by the restatement only."""
import ctypes
import random

import pytest

from alphazeroforhnefatafl_amd import replay
from oracle import oracle as orc


def _rand_record(rng, side, vec_len=None):
    m = [[rng.choice([0, 1, 5, 20, 21, 30, 35, 255]) for _ in range(side)] for _ in range(side)]
    v = [rng.randrange(2) for _ in range(rng.randrange(0, 40) if vec_len is None else vec_len)]
    return m, v, rng.randrange(256), rng.randrange(256)


def _bytes(p):
    with open(p, "rb") as f:
        return f.read()


@pytest.mark.parametrize("side,max_entries", [(7, 1000), (11, 1000), (7, 25), (7, 10), (11, 3), (7, 1), (13, 40)])
def test_append_matches_reference_semantics(tmp_path, side, max_entries):
    rng = random.Random(side * 1000 + max_entries)
    a, b = str(tmp_path / "ours.txt"), str(tmp_path / "ref.txt")
    for i in range(12):
        m, v, x, y = _rand_record(rng, side)
        replay.write_to_file(a, m, v, x, y, max_entries)
        orc.write_to_file(b, m, v, x, y, max_entries)
        assert _bytes(a) == _bytes(b), f"record {i}"


def test_first_record_layout(tmp_path):
    p = str(tmp_path / "one.txt")
    replay.write_to_file(p, [[20, 0, 20], [1, 35, 1], [20, 0, 20]], [1, 0, 1, 1], 7, 200, 100)
    assert _bytes(p) == b"20,0,20\n1,35,1\n20,0,20\n1,0,1,1\n7\n200\n"


def test_line_fifo_drops_one_line_per_call(tmp_path):
    # cap of 8 lines, records of 3+3 = 6 lines: the second call sees 6 < 8 (no drop), the third sees 12 >= 8 and drops ONE line
    p = str(tmp_path / "fifo.txt")
    for k in range(3):
        replay.write_to_file(p, [[k, k, k]] * 3, [k], k, k, 8)
    lines = _bytes(p).decode().split("\n")[:-1]
    assert len(lines) == 6 * 3 - 1 and lines[0] == "0,0,0" and lines[-1] == "2"


def test_existing_file_with_crlf_and_no_final_newline(tmp_path):
    a, b = str(tmp_path / "a.txt"), str(tmp_path / "b.txt")
    for p in (a, b):
        with open(p, "wb") as f:
            f.write(b"1,2\r\n3,4\r\n\r\n9")
    replay.write_to_file(a, [[1, 2], [3, 4]], [], 0, 1, 100)
    orc.write_to_file(b, [[1, 2], [3, 4]], [], 0, 1, 100)
    assert _bytes(a) == _bytes(b)


@pytest.mark.parametrize("max_entries", [1000, 30, 7])
def test_batch_equals_sequential(tmp_path, max_entries):
    rng = random.Random(5)
    side, n = 7, 9
    recs = [_rand_record(rng, side) for _ in range(n)]
    a, b = str(tmp_path / "batch.txt"), str(tmp_path / "seq.txt")
    for p in (a, b):
        replay.write_to_file(p, *recs[0], 1000)
    flat = [x for m, _, _, _ in recs[1:] for row in m for x in row]
    replay.write_records(a, side, flat, [v for _, v, _, _ in recs[1:]], [x for _, _, x, _ in recs[1:]], [y for _, _, _, y in recs[1:]], max_entries)
    for m, v, x, y in recs[1:]:
        orc.write_to_file(b, m, v, x, y, max_entries)
    assert _bytes(a) == _bytes(b)


def test_read_round_trip_and_damaged_head(tmp_path):
    rng = random.Random(11)
    side = 11
    p = str(tmp_path / "rt.txt")
    recs = [_rand_record(rng, side, vec_len=20) for _ in range(6)]
    for m, v, x, y in recs:
        replay.write_to_file(p, m, v, x, y, 10_000)
    got = replay.read(p, side)
    assert got == [(m, v, x, y) for m, v, x, y in recs]
    assert replay.read(p, side, max_records=2) == [(m, v, x, y) for m, v, x, y in recs[-2:]]
    # line-wise FIFO damages only the oldest record: the reader returns the intact tail
    m, v, x, y = _rand_record(rng, side, vec_len=20)
    replay.write_to_file(p, m, v, x, y, 5)
    got = replay.read(p, side)
    assert got[-1] == (m, v, x, y) and got[:-1] == [(a, b, c, d) for a, b, c, d in recs[1:]]


def test_errors(tmp_path):
    from alphazeroforhnefatafl_amd._lib import TaflError
    with pytest.raises(TaflError):
        replay.read(str(tmp_path / "missing.txt"), 7)
    with pytest.raises(TaflError):
        replay.write_to_file(str(tmp_path / "nodir" / "x.txt"), [[1]], [], 0, 0, 10)


@pytest.mark.gpu
def test_write_batch_from_device(tmp_path):
    """Records straight from a GameBatch: matrices by the k_encode_boards kernel, vectors from the k_movegen masks."""
    from alphazeroforhnefatafl_amd import BatchedGameLogic, abi, boards, rules
    n = 5
    lg = BatchedGameLogic(rules.COPENHAGEN, 11)
    b = lg.new_batch(n, boards.COPENHAGEN)
    b.random_advance(3, (ctypes.c_uint32 * n)(*[3 * g for g in range(n)]))
    p = str(tmp_path / "dev.txt")
    replay.write_batch(p, b, [1] * n, [2] * n, 1_000_000)
    got = replay.read(p, 11, vector_cap=lg.action_size)
    assert len(got) == n
    olg = orc.GameLogic(rules.COPENHAGEN, 11)
    states = b.download()
    for g in range(n):
        st = orc.GameState.from_abi(states[g], 128)
        assert got[g][0] == [list(r) for r in st.board_to_matrix()]
        legal = {abi.action_encode(11, pl) for pl in olg.all_plays(st)}
        assert {a for a, bit in enumerate(got[g][1]) if bit} == legal
        assert (got[g][2], got[g][3]) == (1, 2)
