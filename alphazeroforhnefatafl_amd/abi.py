"""ctypes mirror of include/taflhip.h plus the host-side vocabulary of the reference crate.

Every struct here is layout-identical to the C header (checked by tests/test_abi.py against
sizes reported by the built library).  Names follow the reference: Side/PieceType
(game/pieces.rs:13-38), Axis (game/tiles.rs:167-170), Play (game/play.rs:22-68), Tile notation
(game/tiles.rs:137-157), Ruleset + presets (game/rules.rs:83-117, game/preset.rs).
No compute happens in this module.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

ABI_VERSION = 1

# Side — game/pieces.rs:13-16
ATTACKER = 0
DEFENDER = 8

# PieceType — game/pieces.rs:31-38
KING = 0x01
SOLDIER = 0x02
KNIGHT = 0x04
COMMANDER = 0x08
GUARD = 0x10
MERCENARY = 0x20

# Axis — game/tiles.rs:167-170
VERTICAL = 0x00
HORIZONTAL = 0x80

# ThroneRule / KingStrength / KingAttack / EnclosureWinRules — game/rules.rs:5-70
NO_THRONE, NO_PASS, KING_PASS, NO_ENTRY, KING_ENTRY = range(5)
STRONG, STRONG_BY_THRONE, WEAK = range(3)
ARMED, ANVIL, HAMMER = range(3)
ENCL_NONE, WITH_EDGE_ACCESS, WITHOUT_EDGE_ACCESS = range(3)

# PlayInvalid — game/error.rs:49-70 (+1, 0 = valid)
PLAY_OK = 0
PLAY_INVALID_NAMES = ["Ok", "WrongPlayer", "NoPiece", "OutOfBounds", "NoCommonAxis", "BlockedByPiece",
                      "MoveThroughBlockedTile", "MoveOntoBlockedTile", "TooFar", "GameOver"]
(WRONG_PLAYER, NO_PIECE, OUT_OF_BOUNDS, NO_COMMON_AXIS, BLOCKED_BY_PIECE, MOVE_THROUGH_BLOCKED_TILE,
 MOVE_ONTO_BLOCKED_TILE, TOO_FAR, GAME_OVER) = range(1, 10)

# GameStatus / WinReason / DrawReason — game/game/mod.rs:16-70
ONGOING, WIN, DRAW = range(3)
WIN_REASON_NAMES = ["KingEscaped", "ExitFort", "KingCaptured", "AllCaptured", "Enclosed", "NoPlays", "Repetition"]
(KING_ESCAPED, EXIT_FORT, KING_CAPTURED, ALL_CAPTURED, ENCLOSED, WIN_NO_PLAYS, WIN_REPETITION) = range(7)
DRAW_REASON_NAMES = ["Repetition", "NoPlays"]
DRAW_REPETITION, DRAW_NO_PLAYS = range(2)
ROLLOUT_REASON_PLY_CAP = 14
ROLLOUT_REASON_STUCK = 15

MAX_LIMBS = 4

# tafl_mcts_params.flags (include/taflhip.h): tuning fields choose how a search is executed, never what it returns
MCTS_FLAG_FPU_INF = 0x1          # src/mcts.rs:49-51,187: unvisited actions score +inf, new nodes start with visits 1 (oracle-pinned only)
MCTS_PIPELINE_DEFAULT, MCTS_PIPELINE_FUSED, MCTS_PIPELINE_TWO_KERNEL = 0, 1, 2


def mcts_tune(pipeline: int = 0, slots: int = 0, parts: int = 0, share: int = 0) -> int:
    """TAFL_MCTS_TUNE_PIPELINE(pipeline) | TAFL_MCTS_TUNE_SLOTS(slots) | TAFL_MCTS_TUNE_PARTS(parts) | TAFL_MCTS_TUNE_SHARE(share)."""
    return ((pipeline & 15) << 4) | ((slots & 15) << 8) | ((parts & 15) << 12) | ((share & 15) << 16)


def ps_none() -> int:
    return 0x0000


def ps_all() -> int:
    return 0xFFFF


def ps_type(*piece_types: int) -> int:
    """PieceSet::from(Vec<PieceType>) — game/pieces.rs:177-184 (both sides)."""
    v = 0
    for pt in piece_types:
        v |= pt | (pt << 8)
    return v


def ps_piece(piece_type: int, side: int) -> int:
    """PieceSet::from_piece — game/pieces.rs:234-236."""
    return (piece_type << side) & 0xFFFF


def ps_side(side: int) -> int:
    """PieceSet::from(Side) — game/pieces.rs:206-210."""
    return (0xFF << side) & 0xFFFF


class TaflRules(C.Structure):
    _fields_ = [
        ("edge_escape", C.c_uint8), ("king_strength", C.c_uint8), ("king_attack", C.c_uint8),
        ("has_shieldwall", C.c_uint8), ("sw_corners_may_close", C.c_uint8), ("exit_fort", C.c_uint8),
        ("throne_movement", C.c_uint8), ("starting_side", C.c_uint8), ("enclosure_win", C.c_uint8),
        ("has_repetition_rule", C.c_uint8), ("rep_is_loss", C.c_uint8), ("draw_on_no_plays", C.c_uint8),
        ("linnaean_capture", C.c_uint8), ("_pad0", C.c_uint8 * 3),
        ("sw_captures", C.c_uint16), ("may_enter_corners", C.c_uint16), ("hostility_throne", C.c_uint16),
        ("hostility_corners", C.c_uint16), ("hostility_edge", C.c_uint16), ("slow_pieces", C.c_uint16),
        ("n_repetitions", C.c_uint32),
    ]


class TaflPlay(C.Structure):
    _fields_ = [("from_row", C.c_uint8), ("from_col", C.c_uint8), ("axis", C.c_uint8), ("disp", C.c_int8)]


class TaflState(C.Structure):
    _fields_ = [
        ("att", C.c_uint64 * MAX_LIMBS), ("deff", C.c_uint64 * MAX_LIMBS),
        ("turn", C.c_uint32), ("plays_since_capture", C.c_uint32), ("rep_ring", C.c_uint32 * 4),
        ("attacker_reps", C.c_uint16), ("defender_reps", C.c_uint16),
        ("attacker_mid_pair", C.c_uint8), ("defender_mid_pair", C.c_uint8),
        ("side_to_play", C.c_uint8), ("status", C.c_uint8), ("reason", C.c_uint8), ("winner", C.c_uint8),
        ("side_len", C.c_uint8), ("_pad", C.c_uint8),
    ]


class TaflEffects(C.Structure):
    _fields_ = [("captures", C.c_uint64 * MAX_LIMBS), ("code", C.c_uint8), ("status", C.c_uint8),
                ("reason", C.c_uint8), ("winner", C.c_uint8), ("n_captures", C.c_uint8), ("_pad", C.c_uint8 * 3)]


class TaflRolloutResult(C.Structure):
    _fields_ = [("value", C.c_int8), ("status", C.c_uint8), ("reason", C.c_uint8), ("winner", C.c_uint8),
                ("plies", C.c_uint32)]


class TaflRootChild(C.Structure):
    _fields_ = [("play", TaflPlay), ("action", C.c_uint32), ("visits", C.c_uint32), ("q", C.c_double)]


class TaflMctsParams(C.Structure):
    _fields_ = [("n_sims", C.c_uint32), ("max_rollout_plies", C.c_uint32), ("c_puct", C.c_double),
                ("seed", C.c_uint64), ("sim_offset", C.c_uint32), ("flags", C.c_uint32)]


class TaflMctsStats(C.Structure):
    _fields_ = [("sims", C.c_uint64), ("rollouts", C.c_uint64), ("rollout_plies", C.c_uint64),
                ("tree_depth_sum", C.c_uint64), ("children_scanned", C.c_uint64), ("terminal_hits", C.c_uint64),
                ("reason_hist", C.c_uint64 * 16), ("faults", C.c_uint64), ("spec_issued", C.c_uint64),
                ("spec_hits", C.c_uint64)]


class TaflGmctsStats(C.Structure):
    _fields_ = [("sims", C.c_uint64), ("predicts", C.c_uint64), ("terminal_hits", C.c_uint64), ("faults", C.c_uint64),
                ("select_depth_sum", C.c_uint64), ("waiting", C.c_uint64), ("_reserved", C.c_uint64 * 2)]


EXPECTED_SIZES = {"tafl_rules": 32, "tafl_play": 4, "tafl_state": 104, "tafl_effects": 40,
                  "tafl_rollout_result": 8, "tafl_root_child": 24, "tafl_mcts_params": 32,
                  "tafl_mcts_stats": 200, "tafl_gmcts_stats": 64}
for _name, _cls in [("tafl_rules", TaflRules), ("tafl_play", TaflPlay), ("tafl_state", TaflState),
                    ("tafl_effects", TaflEffects), ("tafl_rollout_result", TaflRolloutResult),
                    ("tafl_root_child", TaflRootChild), ("tafl_mcts_params", TaflMctsParams),
                    ("tafl_mcts_stats", TaflMctsStats), ("tafl_gmcts_stats", TaflGmctsStats)]:
    assert C.sizeof(_cls) == EXPECTED_SIZES[_name], (_name, C.sizeof(_cls))


# --- Tile / Play helpers ------------------------------------------------------------------------

class ParseError(ValueError):
    """game/error.rs:6-26."""


class PlayError(ValueError):
    """game/error.rs:36-39 (DisjointTiles)."""


def tile_from_str(s: str) -> tuple[int, int]:
    """Tile::from_str — game/tiles.rs:137-151: 'a8' -> (row 7, col 0)."""
    if not s:
        raise ParseError("EmptyString")
    b = ord(s[0])
    if not (97 <= b <= 122):
        raise ParseError(f"BadChar({s[0]!r})")
    try:
        row = int(s[1:]) - 1
    except ValueError as e:
        raise ParseError(f"BadInt({e})") from None
    if row < 0 or row > 255:
        raise ParseError("BadInt(out of range)")
    return row, b - 97


def tile_to_str(row: int, col: int) -> str:
    """Display for Tile — game/tiles.rs:131-135."""
    return f"{chr(col + 97)}{row + 1}"


def play_from_tiles(src: tuple[int, int], dst: tuple[int, int]) -> TaflPlay:
    """Play::from_tiles — game/play.rs:36-49."""
    if src[0] == dst[0]:
        return TaflPlay(src[0], src[1], HORIZONTAL, dst[1] - src[1])
    if src[1] == dst[1]:
        return TaflPlay(src[0], src[1], VERTICAL, dst[0] - src[0])
    raise PlayError("DisjointTiles")


def play_from_str(s: str) -> TaflPlay:
    """Play::from_str — game/play.rs:70-86: 'a8-a11'."""
    tokens = s.split("-")
    if len(tokens) != 2:
        raise ParseError(f"BadString({s!r})")
    a = tile_from_str(tokens[0])
    b = tile_from_str(tokens[1])
    try:
        return play_from_tiles(a, b)
    except PlayError as e:
        raise ParseError(f"BadPlay({e})") from None


def play_to(p: TaflPlay) -> tuple[int, int]:
    """Play::to — game/play.rs:59-67."""
    if p.axis == VERTICAL:
        return p.from_row + p.disp, p.from_col
    return p.from_row, p.from_col + p.disp


def play_to_str(p: TaflPlay) -> str:
    t = play_to(p)
    return f"{tile_to_str(p.from_row, p.from_col)}-{tile_to_str(*t)}"


def play_tuple(p: TaflPlay) -> tuple[int, int, int, int]:
    """(from_row, from_col, to_row, to_col)."""
    t = play_to(p)
    return (p.from_row, p.from_col, t[0], t[1])


# --- Ruleset ----------------------------------------------------------------------------------

@dataclass(frozen=True)
class Ruleset:
    """game/rules.rs:83-117, field for field."""
    edge_escape: bool = False
    king_strength: int = STRONG
    king_attack: int = ARMED
    shieldwall: tuple[bool, int] | None = None     # (corners_may_close, captures PieceSet)
    exit_fort: bool = False
    throne_movement: int = KING_ENTRY
    may_enter_corners: int = 0
    hostility_throne: int = 0
    hostility_corners: int = 0
    hostility_edge: int = 0
    slow_pieces: int = 0
    starting_side: int = ATTACKER
    enclosure_win: int = ENCL_NONE
    repetition_rule: tuple[int, bool] | None = None  # (n_repetitions, is_loss)
    draw_on_no_plays: bool = False
    linnaean_capture: bool = False

    def replace(self, **kw) -> "Ruleset":
        import dataclasses
        return dataclasses.replace(self, **kw)

    def to_c(self) -> TaflRules:
        r = TaflRules()
        r.edge_escape = int(self.edge_escape)
        r.king_strength = self.king_strength
        r.king_attack = self.king_attack
        r.has_shieldwall = int(self.shieldwall is not None)
        if self.shieldwall is not None:
            r.sw_corners_may_close = int(self.shieldwall[0])
            r.sw_captures = self.shieldwall[1]
        r.exit_fort = int(self.exit_fort)
        r.throne_movement = self.throne_movement
        r.may_enter_corners = self.may_enter_corners
        r.hostility_throne = self.hostility_throne
        r.hostility_corners = self.hostility_corners
        r.hostility_edge = self.hostility_edge
        r.slow_pieces = self.slow_pieces
        r.starting_side = self.starting_side
        r.enclosure_win = self.enclosure_win
        r.has_repetition_rule = int(self.repetition_rule is not None)
        if self.repetition_rule is not None:
            r.n_repetitions = self.repetition_rule[0]
            r.rep_is_loss = int(self.repetition_rule[1])
        r.draw_on_no_plays = int(self.draw_on_no_plays)
        r.linnaean_capture = int(self.linnaean_capture)
        return r

    @staticmethod
    def from_c(r: TaflRules) -> "Ruleset":
        return Ruleset(
            edge_escape=bool(r.edge_escape), king_strength=r.king_strength, king_attack=r.king_attack,
            shieldwall=(bool(r.sw_corners_may_close), r.sw_captures) if r.has_shieldwall else None,
            exit_fort=bool(r.exit_fort), throne_movement=r.throne_movement,
            may_enter_corners=r.may_enter_corners, hostility_throne=r.hostility_throne,
            hostility_corners=r.hostility_corners, hostility_edge=r.hostility_edge,
            slow_pieces=r.slow_pieces, starting_side=r.starting_side, enclosure_win=r.enclosure_win,
            repetition_rule=(r.n_repetitions, bool(r.rep_is_loss)) if r.has_repetition_rule else None,
            draw_on_no_plays=bool(r.draw_on_no_plays), linnaean_capture=bool(r.linnaean_capture))


class rules:
    """game/preset.rs:6-124."""
    COPENHAGEN = Ruleset(
        edge_escape=False, king_strength=STRONG, king_attack=ARMED,
        shieldwall=(True, ps_type(SOLDIER)), exit_fort=True, throne_movement=KING_ENTRY,
        may_enter_corners=ps_type(KING), hostility_throne=ps_all(), hostility_corners=ps_type(SOLDIER),
        hostility_edge=ps_none(), slow_pieces=ps_none(), starting_side=ATTACKER,
        enclosure_win=WITHOUT_EDGE_ACCESS, repetition_rule=(3, True), draw_on_no_plays=False,
        linnaean_capture=False)
    BRANDUBH = Ruleset(
        edge_escape=False, king_strength=STRONG_BY_THRONE, king_attack=ARMED, shieldwall=None,
        exit_fort=False, throne_movement=KING_ENTRY, may_enter_corners=ps_type(KING),
        hostility_throne=ps_type(SOLDIER), hostility_corners=ps_all(), hostility_edge=ps_none(),
        slow_pieces=ps_none(), starting_side=ATTACKER, enclosure_win=WITHOUT_EDGE_ACCESS,
        repetition_rule=(3, True), draw_on_no_plays=False, linnaean_capture=False)
    MAGPIE = Ruleset(
        edge_escape=False, king_strength=STRONG, king_attack=ARMED, shieldwall=None, exit_fort=False,
        throne_movement=KING_ENTRY, may_enter_corners=ps_type(KING), hostility_throne=ps_all(),
        hostility_corners=ps_all(), hostility_edge=ps_none(), slow_pieces=ps_type(KING),
        starting_side=ATTACKER, enclosure_win=ENCL_NONE, repetition_rule=None, draw_on_no_plays=False,
        linnaean_capture=False)
    TABLUT = Ruleset(
        edge_escape=True, king_strength=STRONG_BY_THRONE, king_attack=ARMED, shieldwall=None,
        exit_fort=False, throne_movement=NO_ENTRY, may_enter_corners=ps_all(), hostility_throne=ps_all(),
        hostility_corners=ps_none(), hostility_edge=ps_none(), slow_pieces=ps_none(),
        starting_side=ATTACKER, enclosure_win=ENCL_NONE, repetition_rule=(3, False),
        draw_on_no_plays=True, linnaean_capture=True)
    KOCH = Ruleset(
        edge_escape=False, king_strength=STRONG_BY_THRONE, king_attack=ARMED, shieldwall=None,
        exit_fort=False, throne_movement=KING_ENTRY, may_enter_corners=ps_type(KING),
        hostility_throne=ps_all(), hostility_corners=ps_type(SOLDIER), hostility_edge=ps_none(),
        slow_pieces=ps_none(), starting_side=ATTACKER, enclosure_win=WITHOUT_EDGE_ACCESS,
        repetition_rule=(3, True), draw_on_no_plays=False, linnaean_capture=False)

    BY_NAME = {}


rules.BY_NAME = {"copenhagen": rules.COPENHAGEN, "brandubh": rules.BRANDUBH, "magpie": rules.MAGPIE,
                 "tablut": rules.TABLUT, "koch": rules.KOCH}


class boards:
    """game/preset.rs:126-135 (+ the build-defined 13x13 start)."""
    COPENHAGEN = "3ttttt3/5t5/11/t4T4t/t3TTT3t/tt1TTKTT1tt/t3TTT3t/t4T4t/11/5t5/3ttttt3"
    BRANDUBH = "3t3/3t3/3T3/ttTKTtt/3T3/3t3/3t3"
    MAGPIE = "3t3/1t3t1/3T3/t1TKT1t/3T3/1t3t1/3t3"
    TABLUT = "3ttt3/4t4/4T4/t3T3t/ttTTKTTtt/t3T3t/4T4/4t4/3ttt3"
    COPENHAGEN13 = "4ttttt4/6t6/13/13/t5T5t/t4TTT4t/tt2TTKTT2tt/t4TTT4t/t5T5t/13/13/6t6/4ttttt4"


def row_width(word_bits: int) -> int:
    """BitField::ROW_WIDTH — game/bitfield.rs:178-181."""
    return {64: 7, 128: 11, 256: 15, 512: 21}[word_bits]


def word_bits_for(side_len: int) -> int:
    """Smallest reference board word that carries the board (game/board/state.rs:332-340)."""
    if side_len <= 7:
        return 64
    if side_len <= 11:
        return 128
    if side_len <= 15:
        return 256
    raise ValueError("side_len > 15 is not supported (king nibble, SURVEY.md §8a2)")


def fen_side_len(fen: str) -> int:
    """side_len is inferred from the first FEN row — game/board/state.rs:225-250."""
    n, run = 0, 0
    for ch in fen.split("/")[0]:
        if ch.isdigit():
            run = run * 10 + int(ch)
        else:
            n += run + 1
            run = 0
    return n + run


def state_words(st: TaflState, word_bits: int) -> tuple[int, int]:
    """(attackers, defenders) as Python ints in the reference layout."""
    k = word_bits // 64
    a = sum(int(st.att[i]) << (64 * i) for i in range(k))
    d = sum(int(st.deff[i]) << (64 * i) for i in range(k))
    return a, d


def action_size(side_len: int) -> int:
    """n^2 * 2(n-1): every (from tile, destination on its row/column) pair — include/taflhip.h."""
    return side_len * side_len * 2 * (side_len - 1)


def action_encode(side_len: int, p: TaflPlay) -> int:
    n, m = side_len, side_len - 1
    r, c, dist = p.from_row, p.from_col, abs(p.disp)
    if p.axis == VERTICAL:
        slot = dist - 1 if p.disp > 0 else (m - r) + dist - 1
    else:
        slot = m + dist - 1 if p.disp > 0 else m + (m - c) + dist - 1
    return (r * n + c) * 2 * m + slot


def action_decode(side_len: int, a: int) -> TaflPlay:
    n, m = side_len, side_len - 1
    sq, slot = divmod(a, 2 * m)
    r, c = divmod(sq, n)
    if slot < m - r:
        return TaflPlay(r, c, VERTICAL, slot + 1)
    if slot < m:
        return TaflPlay(r, c, VERTICAL, -(slot - (m - r) + 1))
    if slot < m + (m - c):
        return TaflPlay(r, c, HORIZONTAL, slot - m + 1)
    return TaflPlay(r, c, HORIZONTAL, -(slot - m - (m - c) + 1))


def state_to_fen(st: TaflState, word_bits: int) -> str:
    """BoardState::to_fen (game/board/state.rs:271-295) through the library's host helper (no GPU needed)."""
    from ._lib import check, lib
    buf = C.create_string_buffer(1024)
    n = lib().tafl_state_to_fen(C.byref(st), word_bits, buf, 1024)
    if n < 0:
        check(n)
    return buf.value.decode()
