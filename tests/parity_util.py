"""Shared helpers for differential tests (oracle vs host-sim vs GPU): workloads and comparisons."""
import ctypes as C
import random

from alphazeroforhnefatafl_amd import abi
from alphazeroforhnefatafl_amd.abi import TaflPlay, TaflState

CONFIGS = {
    # name: (rules, start FEN, word_bits)
    "copenhagen11": (abi.rules.COPENHAGEN, abi.boards.COPENHAGEN, 128),
    "brandubh7": (abi.rules.BRANDUBH, abi.boards.BRANDUBH, 64),
    "brandubh7_u128": (abi.rules.BRANDUBH, abi.boards.BRANDUBH, 128),
    "tablut9": (abi.rules.TABLUT, abi.boards.TABLUT, 128),
    "magpie7": (abi.rules.MAGPIE, abi.boards.MAGPIE, 64),
    "koch7": (abi.rules.KOCH, abi.boards.BRANDUBH, 64),
    "copenhagen13": (abi.rules.COPENHAGEN, abi.boards.COPENHAGEN13, 256),
    "copenhagen9_u256": (abi.rules.COPENHAGEN, abi.boards.TABLUT, 256),
}


def clone_states(states, n):
    out = (TaflState * n)()
    C.memmove(out, states, C.sizeof(TaflState) * n)
    return out


def states_equal(a, b, n):
    return bytes(a)[: C.sizeof(TaflState) * n] == bytes(b)[: C.sizeof(TaflState) * n]


def first_state_diff(a, b, n):
    sz = C.sizeof(TaflState)
    ba, bb = bytes(a), bytes(b)
    for i in range(n):
        if ba[i * sz:(i + 1) * sz] != bb[i * sz:(i + 1) * sz]:
            return i
    return -1


def describe_state(st: TaflState, word_bits: int) -> str:
    rw = abi.row_width(word_bits)
    a, d = abi.state_words(st, word_bits)
    bits = word_bits
    krow, kcol = (d >> (bits - 4)) & 15, (a >> (bits - 4)) & 15
    rows = []
    for r in range(st.side_len):
        row = ""
        for c in range(st.side_len):
            i = r * rw + c
            if (d >> i) & 1:
                row += "K" if (r, c) == (krow, kcol) else "T"
            elif (a >> i) & 1:
                row += "t"
            else:
                row += "."
        rows.append(row)
    return "\n".join(rows) + f"\nside={st.side_to_play} status={st.status}/{st.reason}/{st.winner} turn={st.turn} " \
                             f"reps={st.attacker_reps},{st.defender_reps} king=({krow},{kcol})"


def start_states(oracle_mod, fen, side, word_bits, n):
    st = oracle_mod.GameState(fen, side, word_bits).to_abi()
    arr = (TaflState * n)()
    for i in range(n):
        C.memmove(C.byref(arr, i * C.sizeof(TaflState)), C.byref(st), C.sizeof(TaflState))
    return arr


def random_board_states(rng: random.Random, side_len, word_bits, n, density=0.35, with_king=True):
    """Synthetic (not necessarily reachable) positions: stresses enclosure / shieldwall / exit-fort paths."""
    rw = abi.row_width(word_bits)
    arr = (TaflState * n)()
    k = word_bits // 64
    for g in range(n):
        att = deff = 0
        dens = rng.choice([0.15, density, 0.5, 0.7])
        ratio = rng.choice([0.3, 0.5, 0.7])
        cells = [(r, c) for r in range(side_len) for c in range(side_len)]
        occupied = []
        for (r, c) in cells:
            if rng.random() < dens:
                if rng.random() < ratio:
                    att |= 1 << (r * rw + c)
                else:
                    deff |= 1 << (r * rw + c)
                    occupied.append((r, c))
        kr = kc = 0
        if with_king and rng.random() < 0.95:
            if occupied and rng.random() < 0.9:
                kr, kc = rng.choice(occupied)
            else:
                kr, kc = rng.randrange(side_len), rng.randrange(side_len)
                att &= ~(1 << (kr * rw + kc))
                deff |= 1 << (kr * rw + kc)
        att |= kc << (word_bits - 4)
        deff |= kr << (word_bits - 4)
        st = arr[g]
        for i in range(k):
            st.att[i] = (att >> (64 * i)) & 0xFFFFFFFFFFFFFFFF
            st.deff[i] = (deff >> (64 * i)) & 0xFFFFFFFFFFFFFFFF
        st.side_to_play = rng.choice([abi.ATTACKER, abi.DEFENDER])
        st.side_len = side_len
        st.status = abi.ONGOING
    return arr


def random_plays(rng: random.Random, side_len, n):
    plays = (TaflPlay * n)()
    for g in range(n):
        plays[g] = TaflPlay(rng.randrange(side_len + 1), rng.randrange(side_len + 1),
                            rng.choice([abi.VERTICAL, abi.HORIZONTAL]),
                            rng.choice([-1, 1]) * rng.randrange(1, side_len + 1))
    return plays


def effects_tuple(e):
    return (tuple(int(x) for x in e.captures), e.code, e.status, e.reason, e.winner, e.n_captures)


def play_tuple4(p):
    return (p.from_row, p.from_col, p.axis, p.disp)
