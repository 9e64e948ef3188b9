"""The reference's text replay buffer (`write_to_file`, game/main.rs:86-132) — host mirror over the C-ABI.

`write_to_file(path, matrix, vector, value1, value2, max_entries)` keeps the reference's name, argument order and its
line-wise FIFO rule (see include/taflhip.h); `write_batch` appends one record per game of a `GameBatch`
(board_to_matrix planes from the device + the dense legal-move vector), `read` returns the newest complete records.
"""
from __future__ import annotations

import ctypes as C

from ._lib import check, lib


def _u8(seq):
    return (C.c_uint8 * len(seq))(*seq)


def write_to_file(file_path: str, matrix, vector, value1: int, value2: int, max_entries: int) -> None:
    """game/main.rs:86-132 for one record; `matrix` is side_len rows of side_len u8 values."""
    n = len(matrix)
    flat = [int(v) for row in matrix for v in row]
    if len(flat) != n * n:
        raise ValueError("matrix must be square")
    vec = [int(v) for v in vector]
    check(lib().tafl_replay_append(file_path.encode(), _u8(flat), n, _u8(vec) if vec else None, len(vec), value1, value2, max_entries))


def write_records(file_path: str, side_len: int, matrices, vectors, values1, values2, max_entries: int) -> None:
    """n records = n consecutive write_to_file calls (one read / one write of the file).  `matrices`: flat u8 buffer or list
    of n*side_len*side_len values; `vectors`: list of n u8 sequences."""
    n = len(vectors)
    offs = [0]
    for v in vectors:
        offs.append(offs[-1] + len(v))
    flatv = [int(x) for v in vectors for x in v]
    mats = matrices if isinstance(matrices, C.Array) else _u8([int(x) for x in matrices])
    if len(mats) != n * side_len * side_len:
        raise ValueError("matrices must hold n * side_len * side_len values")
    check(lib().tafl_replay_append_batch(file_path.encode(), C.cast(mats, C.POINTER(C.c_uint8)), side_len, n, _u8(flatv) if flatv else None,
                                         (C.c_uint32 * (n + 1))(*offs), _u8([int(x) for x in values1]), _u8([int(x) for x in values2]), max_entries))


def write_batch(file_path: str, batch, values1, values2, max_entries: int) -> None:
    """One record per game of `batch`: matrix = board_to_matrix (tafl_encode_boards, device kernel), vector = the dense
    legal-move vector of the side to move (1 = legal, length action_size; the role of validate_moves, game/main.rs:45-52)."""
    n, side, asz, mw = batch.n, batch.logic.side_len, batch.logic.action_size, batch.logic.mask_words
    mats = batch.encode_boards()
    _, masks = batch.iter_plays(want_masks=True)
    vectors = []
    for g in range(n):
        words = masks[g * mw:(g + 1) * mw]
        vectors.append([(words[a >> 5] >> (a & 31)) & 1 for a in range(asz)])
    write_records(file_path, side, mats, vectors, values1, values2, max_entries)


def read(file_path: str, side_len: int, max_records: int = 1 << 20, vector_cap: int = 8192):
    """Newest complete records, oldest first: list of (matrix rows, vector, value1, value2)."""
    L = lib()
    n = C.c_uint32()
    check(L.tafl_replay_read(file_path.encode(), side_len, max_records, None, None, 0, None, None, None, C.byref(n)))
    k = n.value
    mats = (C.c_uint8 * max(1, k * side_len * side_len))()
    vecs = (C.c_uint8 * max(1, k * vector_cap))()
    lens = (C.c_uint32 * max(1, k))()
    v1 = (C.c_uint8 * max(1, k))()
    v2 = (C.c_uint8 * max(1, k))()
    check(L.tafl_replay_read(file_path.encode(), side_len, k, mats, vecs, vector_cap, lens, v1, v2, C.byref(n)))
    out = []
    for r in range(n.value):
        m = [list(mats[(r * side_len + i) * side_len:(r * side_len + i + 1) * side_len]) for i in range(side_len)]
        out.append((m, list(vecs[r * vector_cap:r * vector_cap + lens[r]]), v1[r], v2[r]))
    return out
