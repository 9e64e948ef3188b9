"""Deterministic stand-in for nnet.predict (src/mcts.py:85), shared by the golden generator (which feeds it to the
reference's own mcts.py), the oracle tests and the device tests.

Input: the board_to_matrix planes (game/main.rs:55-83) of the position + the side to move.  Output: (priors float32
[action_size], value float32).  No transcendental functions (their last bit may differ between machines): priors are
24-bit mantissas scaled by powers of two, so that their float64 sum DOES round and the order of np.sum matters; about one
position in seven gets all-zero priors (the "all valid moves were masked" branch, mcts.py:91-98), one in five a sparse prior.
"""
import zlib

import numpy as np


def stub_predict(matrix_bytes: bytes, side: int, action_size: int, salt: int = 0):
    seed = zlib.crc32(bytes(matrix_bytes) + bytes([side & 255, salt & 255]))
    rs = np.random.RandomState(seed)
    mant = rs.randint(1 << 23, 1 << 24, size=action_size).astype(np.float32)
    expo = rs.randint(0, 40, size=action_size)
    pri = np.ldexp(mant, -24 - expo).astype(np.float32)
    kind = seed % 35
    if kind % 7 == 0:
        pri[:] = 0
    elif kind % 5 == 0:
        pri[rs.randint(0, 2, size=action_size) == 0] = 0
    value = np.float32(rs.randint(-(1 << 20), (1 << 20) + 1)) / np.float32(1 << 20)
    return pri, np.float32(value)


def matrix_bytes_of(rows) -> bytes:
    return bytes(int(v) for r in rows for v in r)
