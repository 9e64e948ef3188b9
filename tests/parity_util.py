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
    # a 13x13 board under rules that are NOT the Copenhagen preset: the run-time-rules kernels of the 256-bit word (8 limbs, 15 columns)
    "tablut13_u256": (abi.rules.TABLUT, abi.boards.COPENHAGEN13, 256),
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


# ---- crafted workloads for the rare rules (enclosure win, exit fort, shieldwall, repetition, no plays) -----------

def _blank(side_len, word_bits):
    return {"n": side_len, "wb": word_bits, "att": set(), "def": set(), "king": None}


def _to_state(b, side) -> TaflState:
    rw = abi.row_width(b["wb"])
    att = deff = 0
    for (r, c) in b["att"]:
        att |= 1 << (r * rw + c)
    for (r, c) in b["def"]:
        deff |= 1 << (r * rw + c)
    kr, kc = b["king"] if b["king"] else (0, 0)
    if b["king"]:
        deff |= 1 << (kr * rw + kc)
    att |= kc << (b["wb"] - 4)
    deff |= kr << (b["wb"] - 4)
    st = TaflState()
    for i in range(b["wb"] // 64):
        st.att[i] = (att >> (64 * i)) & 0xFFFFFFFFFFFFFFFF
        st.deff[i] = (deff >> (64 * i)) & 0xFFFFFFFFFFFFFFFF
    st.side_to_play = side
    st.side_len = b["n"]
    return st


def _transform(n, r, c, k):
    """k in 0..7: the 8 symmetries of the square."""
    if k & 4:
        r, c = c, r
    if k & 1:
        r = n - 1 - r
    if k & 2:
        c = n - 1 - c
    return r, c


def enclosure_positions(rng, side_len, word_bits, count):
    """Attackers ring a region holding (mostly) all defenders; one ring tile is open and an attacker can close it."""
    out = []
    n = side_len
    while len(out) < count:
        b = _blank(n, word_bits)
        touch_edge = rng.random() < 0.25
        lo = 0 if touch_edge else 1
        r0 = rng.randrange(lo, n - 3); r1 = rng.randrange(r0 + 2, min(n - lo, r0 + 7))
        c0 = rng.randrange(lo, n - 3); c1 = rng.randrange(c0 + 2, min(n - lo, c0 + 7))
        ring = [(r, c) for r in range(r0, r1 + 1) for c in range(c0, c1 + 1)
                if (r in (r0, r1) or c in (c0, c1)) and not ((r in (r0, r1)) and (c in (c0, c1)))]
        inner = [(r, c) for r in range(r0 + 1, r1) for c in range(c0 + 1, c1)]
        if not inner or len(ring) < 4:
            continue
        b["att"] = set(ring)
        if rng.random() < 0.3:   # ring corners sometimes filled
            b["att"] |= {(r0, c0), (r0, c1), (r1, c0), (r1, c1)}
        b["king"] = rng.choice(inner)
        for t in inner:
            if t != b["king"] and rng.random() < 0.4:
                b["def"].add(t)
        if rng.random() < 0.15:   # a few stray pieces inside/outside make negatives
            b["att"].add(rng.choice(inner)) if rng.random() < 0.5 else None
            b["att"].discard(b["king"])
        gap = rng.choice(ring)
        b["att"].discard(gap)
        # attacker that can close the gap: along the outward normal, or anywhere in the row/col with clear path
        gr, gc = gap
        cand = []
        for dr, dc in ((-1, 0), (1, 0), (0, -1), (0, 1)):
            rr, cc = gr + dr, gc + dc
            steps = 0
            while 0 <= rr < n and 0 <= cc < n and (rr, cc) not in b["att"] and (rr, cc) not in b["def"] and (rr, cc) != b["king"]:
                if not (r0 <= rr <= r1 and c0 <= cc <= c1):
                    cand.append((rr, cc))
                rr += dr; cc += dc; steps += 1
        if not cand:
            continue
        b["att"].add(rng.choice(cand))
        # outside clutter
        for _ in range(rng.randrange(0, 6)):
            t = (rng.randrange(n), rng.randrange(n))
            if not (r0 <= t[0] <= r1 and c0 <= t[1] <= c1) and t not in b["att"] and t != gap:
                if rng.random() < 0.85:
                    b["att"].add(t)
                else:
                    b["def"].add(t)
        b["def"].discard(b["king"])
        b["def"] -= b["att"]
        out.append(_to_state(b, abi.ATTACKER))
    return out


_FORT_SEEDS_9 = ["9/9/8t/7tT/7T1/6tT1/7TK/7tT/9", "9/9/9/9/9/5TTTT/5T2K/6TTT/9", "9/9/9/8T/7Tt/7T1/7TK/8T/9",
                 "9/9/9/8T/7TT/7TT/7TK/8T/9", "9/9/9/8T/9/4t2T1/7TK/8T/9", "9/9/9/9/9/6TTT/5T2K/6TTT/9"]


def _fen_cells(fen):
    cells = {}
    for r, line in enumerate(fen.split("/")):
        c = 0
        run = 0
        for ch in line:
            if ch.isdigit():
                run = run * 10 + int(ch)
            else:
                c += run
                run = 0
                cells[(r, c)] = ch
                c += 1
    return cells


def exit_fort_positions(rng, side_len, word_bits, count):
    """Variations of the reference's exit-fort shapes (logic.rs:1217-1222) under the 8 symmetries, shifted and perturbed,
    plus one free defender far away so that every defender move re-evaluates the fort."""
    out = []
    n = side_len
    while len(out) < count:
        seed = _fen_cells(rng.choice(_FORT_SEEDS_9))
        k = rng.randrange(8)
        shift = rng.randrange(-2, n - 9 + 3)
        b = _blank(n, word_bits)
        ok = True
        for (r, c), ch in seed.items():
            # seeds sit on the right edge (col 8 of 9): move them to this board's right edge, then shift along it
            rr, cc = r + shift, c + (n - 9)
            if not (0 <= rr < n):
                ok = False
                break
            rr, cc = _transform(n, rr, cc, k)
            if ch == "K":
                b["king"] = (rr, cc)
            elif ch == "T":
                b["def"].add((rr, cc))
            else:
                b["att"].add((rr, cc))
        if not ok or b["king"] is None:
            continue
        # perturb
        for _ in range(rng.choice([0, 0, 1, 2])):
            t = (rng.randrange(n), rng.randrange(n))
            if t == b["king"]:
                continue
            mode = rng.random()
            if mode < 0.4:
                b["def"].discard(t); b["att"].discard(t)
            elif mode < 0.7:
                b["att"].discard(t); b["def"].add(t)
            else:
                b["def"].discard(t); b["att"].add(t)
        # a free defender and a free attacker somewhere
        for which in ("def", "att"):
            for _ in range(10):
                t = (rng.randrange(n), rng.randrange(n))
                if t != b["king"] and t not in b["def"] and t not in b["att"]:
                    b[which].add(t)
                    break
        out.append(_to_state(b, abi.DEFENDER))
    return out


def shieldwall_positions(rng, side_len, word_bits, count):
    """A line of pieces on an edge pinned from the inner row; the mover can bracket it (logic.rs:471-569)."""
    out = []
    n = side_len
    while len(out) < count:
        b = _blank(n, word_bits)
        mover = rng.choice([abi.ATTACKER, abi.DEFENDER])
        mine, theirs = ("att", "def") if mover == abi.ATTACKER else ("def", "att")
        L = rng.randrange(1, 5)
        start = rng.randrange(0, n - L - 1)       # wall occupies cols start+1 .. start+L on row 0
        wall = [(0, c) for c in range(start + 1, start + L + 1)]
        for t in wall:
            b[theirs].add(t)
            if rng.random() < 0.9:
                b[mine].add((1, t[1]))
            elif rng.random() < 0.5:
                b[theirs].add((1, t[1]))
        left, right = (0, start), (0, start + L + 1)
        closed_left = rng.random() < 0.5
        fixed, arrive = (left, right) if closed_left else (right, left)
        if rng.random() < 0.85 and 0 <= fixed[1] < n:
            if fixed[1] in (0, n - 1) and rng.random() < 0.5:
                pass                                   # a (closing) corner
            else:
                b[mine].add(fixed)
        king_in_wall = mover == abi.ATTACKER and rng.random() < 0.3
        if king_in_wall:
            kt = rng.choice(wall)
            b["def"].discard(kt)
            b["king"] = kt
        elif mover == abi.DEFENDER or rng.random() < 0.7:
            for _ in range(20):
                t = (rng.randrange(2, n), rng.randrange(n))
                if t not in b["att"] and t not in b["def"]:
                    b["king"] = t
                    break
        # the arriving piece: somewhere on the column of `arrive` (clear path) or along row 0
        if 0 <= arrive[1] < n and arrive not in b["att"] and arrive not in b["def"]:
            d = rng.randrange(2, n)
            t = (d, arrive[1])
            if t not in b["att"] and t not in b["def"] and t != b["king"]:
                b[mine].add(t)
                for rr in range(1, d):
                    b["att"].discard((rr, arrive[1])); b["def"].discard((rr, arrive[1]))
                    if b["king"] == (rr, arrive[1]):
                        b["king"] = None
        b["def"].discard(b["king"]) if b["king"] else None
        b["att"] -= b["def"]
        if b["king"]:
            b["att"].discard(b["king"])
        k = rng.randrange(8)
        tb = _blank(n, word_bits)
        tb["att"] = {_transform(n, r, c, k) for (r, c) in b["att"]}
        tb["def"] = {_transform(n, r, c, k) for (r, c) in b["def"]}
        tb["king"] = _transform(n, *b["king"], k) if b["king"] else None
        out.append(_to_state(tb, mover))
    return out


def sparse_endgame_positions(rng, side_len, word_bits, count):
    """Few defenders, many attackers (and vice versa): AllCaptured / NoPlays / king captures."""
    out = []
    n = side_len
    while len(out) < count:
        b = _blank(n, word_bits)
        kr, kc = rng.randrange(n), rng.randrange(n)
        b["king"] = (kr, kc)
        for dr, dc in ((-1, 0), (1, 0), (0, -1), (0, 1), (-2, 0), (2, 0), (0, -2), (0, 2), (1, 1), (-1, -1), (1, -1), (-1, 1)):
            t = (kr + dr, kc + dc)
            if 0 <= t[0] < n and 0 <= t[1] < n and rng.random() < 0.55:
                b["att"].add(t)
        for _ in range(rng.randrange(0, 3)):
            t = (rng.randrange(n), rng.randrange(n))
            if t != b["king"] and t not in b["att"]:
                b["def"].add(t)
        for _ in range(rng.randrange(0, 8)):
            t = (rng.randrange(n), rng.randrange(n))
            if t != b["king"] and t not in b["def"]:
                b["att"].add(t)
        if rng.random() < 0.15:
            b["king"] = None
        out.append(_to_state(b, rng.choice([abi.ATTACKER, abi.ATTACKER, abi.DEFENDER])))
    return out


def states_array(lst):
    arr = (TaflState * len(lst))()
    for i, s in enumerate(lst):
        C.memmove(C.byref(arr, i * C.sizeof(TaflState)), C.byref(s), C.sizeof(TaflState))
    return arr


def expand_all(states, n, counts):
    """Replicates state g counts[g] times with ranks 0..counts[g]-1 (exhaustive one-ply expansion)."""
    total = sum(int(c) for c in counts)
    arr = (TaflState * total)()
    ranks = (C.c_uint32 * total)()
    src = []
    i = 0
    sz = C.sizeof(TaflState)
    for g in range(n):
        for k in range(int(counts[g])):
            C.memmove(C.byref(arr, i * sz), C.byref(states, g * sz), sz)
            ranks[i] = k
            src.append(g)
            i += 1
    return arr, ranks, total, src


def random_ruleset(rng: random.Random) -> abi.Ruleset:
    def ps():
        return rng.choice([abi.ps_none(), abi.ps_all(), abi.ps_type(abi.KING), abi.ps_type(abi.SOLDIER),
                           abi.ps_piece(abi.SOLDIER, abi.ATTACKER), abi.ps_piece(abi.SOLDIER, abi.DEFENDER),
                           abi.ps_side(abi.DEFENDER), abi.ps_side(abi.ATTACKER)])
    return abi.Ruleset(
        edge_escape=rng.random() < 0.3, king_strength=rng.randrange(3), king_attack=rng.randrange(3),
        shieldwall=None if rng.random() < 0.4 else (rng.random() < 0.5, ps()),
        exit_fort=rng.random() < 0.5, throne_movement=rng.randrange(5), may_enter_corners=ps(),
        hostility_throne=ps(), hostility_corners=ps(), hostility_edge=ps(), slow_pieces=rng.choice([0, 0, ps()]),
        starting_side=abi.ATTACKER, enclosure_win=rng.randrange(3),
        repetition_rule=None if rng.random() < 0.3 else (rng.randrange(1, 4), rng.random() < 0.5),
        draw_on_no_plays=rng.random() < 0.5, linnaean_capture=rng.random() < 0.5)
