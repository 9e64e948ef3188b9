"""ctypes front-end of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under alphazeroforhnefatafl_amd/ does.  The classes mirror the reference's single-game
API (GameLogic / GameState / Game, game/game/logic.rs, game/game/state.rs, game/game/mod.rs) so
that the KAT replays in tests/test_oracle_kat.py read like the reference's own unit tests.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))

from alphazeroforhnefatafl_amd import abi  # noqa: E402  (ABI types only; no compute)
from alphazeroforhnefatafl_amd.abi import (TaflEffects, TaflMctsParams, TaflMctsStats, TaflPlay,  # noqa: E402
                                          TaflRolloutResult, TaflRootChild, TaflRules, TaflState)

_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, "tafl_oracle.c"), os.path.join(_HERE, "tafl_oracle.h"),
           os.path.join(_HERE, "..", "include", "taflhip.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, u8, u16, u32, u64, i32 = C.c_void_p, C.c_uint8, C.c_uint16, C.c_uint32, C.c_uint64, C.c_int
        P = C.POINTER

        def sig(name, res, *args):
            f = getattr(L, name)
            f.restype = res
            f.argtypes = list(args)

        for n in ("orc_sizeof_state", "orc_sizeof_logic", "orc_sizeof_enclosure", "orc_sizeof_tracker"):
            sig(n, C.c_size_t)
        sig("orc_logic_init", i32, vp, P(TaflRules), u8)
        sig("orc_state_init", i32, vp, C.c_char_p, u8, u32)
        sig("orc_state_export", i32, vp, P(TaflState))
        sig("orc_state_import", i32, vp, P(TaflState), u32)
        sig("orc_preset_rules", i32, C.c_char_p, P(TaflRules))
        sig("orc_preset_board", C.c_char_p, C.c_char_p)
        sig("orc_get_piece", i32, vp, u8, u8)
        sig("orc_set_piece", None, vp, u8, u8, u8, u8)
        sig("orc_clear_tile", None, vp, u8, u8)
        sig("orc_move_piece", i32, vp, u8, u8, u8, u8)
        sig("orc_swap_pieces", None, vp, u8, u8, u8, u8)
        sig("orc_get_king", i32, vp)
        sig("orc_tile_occupied", i32, vp, u8, u8)
        sig("orc_count_pieces", i32, vp, u8)
        sig("orc_iter_occupied", i32, vp, u8, P(u8), i32)
        sig("orc_to_fen", i32, vp, C.c_char_p, i32)
        sig("orc_from_display_str", i32, vp, C.c_char_p, u32)
        sig("orc_board_to_matrix", i32, vp, P(u8))
        sig("orc_neighbors", i32, vp, u8, u8, P(u8))
        sig("orc_tiles_between", i32, vp, u8, u8, u8, u8, P(u8))
        sig("orc_validate_play", i32, vp, vp, TaflPlay)
        sig("orc_validate_play_for_side", i32, vp, vp, TaflPlay, u8)
        sig("orc_iter_plays", i32, vp, vp, u8, u8, P(TaflPlay), i32)
        sig("orc_all_plays", i32, vp, vp, P(TaflPlay), i32)
        sig("orc_side_can_play", i32, vp, vp, u8)
        sig("orc_get_captures", i32, vp, vp, TaflPlay, u8, u8, P(u8), i32)
        sig("orc_detect_shieldwall", i32, vp, vp, TaflPlay, P(u8), i32)
        sig("orc_find_enclosure", i32, vp, vp, u8, u8, u16, u16, i32, i32, vp)
        sig("orc_enclosure_tiles", i32, vp, i32, P(u8), i32)
        sig("orc_enclosure_secure", i32, vp, vp, vp, i32, i32)
        sig("orc_detect_exit_fort", i32, vp, vp)
        sig("orc_do_play", i32, vp, vp, TaflPlay, P(TaflEffects))
        sig("orc_do_valid_play", i32, vp, vp, TaflPlay, P(TaflEffects))
        sig("orc_tracker_init", None, vp)
        sig("orc_tracker_track_play", None, vp, u8, TaflPlay, i32)
        sig("orc_tracker_get_repetitions", C.c_size_t, vp, u8)
        sig("orc_action_size", u32, u8)
        sig("orc_action_encode", u32, u8, TaflPlay)
        sig("orc_action_decode", TaflPlay, u8, u32)
        sig("orc_rng", u32, u64, u64, u32, u32)
        sig("orc_rollout_order_plays", i32, vp, vp, P(TaflPlay), i32)
        sig("orc_rollout", i32, vp, vp, u64, u64, u32, u32, P(TaflRolloutResult))
        sig("orc_state_hash", u32, vp)
        sig("orc_random_advance", i32, vp, vp, u64, u64, u32)
        L._predict_t = C.CFUNCTYPE(None, vp, vp, P(C.c_float), P(C.c_float))
        sig("orc_gmcts_new", vp, vp, vp, C.c_double, L._predict_t, vp)
        sig("orc_gmcts_free", None, vp)
        sig("orc_gmcts_run", i32, vp, u32)
        sig("orc_gmcts_root_children", i32, vp, P(TaflRootChild), i32)
        sig("orc_gmcts_root_ns", u32, vp)
        sig("orc_gmcts_root_priors", i32, vp, P(C.c_double))
        sig("orc_gmcts_counts", None, vp, P(u64))
        sig("orc_np_sum", C.c_double, P(C.c_double), C.c_long)
        sig("orc_mcts_new", vp, vp, vp, P(TaflMctsParams), u64)
        sig("orc_mcts_free", None, vp)
        sig("orc_mcts_run", i32, vp)
        sig("orc_mcts_root_children", i32, vp, P(TaflRootChild), i32)
        sig("orc_mcts_root_ns", u32, vp)
        sig("orc_mcts_get_stats", None, vp, P(TaflMctsStats))
        sig("orc_batch_movegen", i32, vp, P(TaflState), u32, u32, P(u32), P(u32), u32)
        sig("orc_batch_step", i32, vp, P(TaflState), u32, u32, P(TaflPlay), P(TaflEffects))
        sig("orc_batch_step_kth", i32, vp, P(TaflState), u32, u32, P(u32), P(TaflPlay), P(TaflEffects))
        sig("orc_batch_rollout", i32, vp, P(TaflState), u32, u32, u64, u32, u32, u64, P(TaflRolloutResult))
        sig("orc_batch_random_advance", i32, vp, P(TaflState), u32, u32, u64, P(u32), u64)
        sig("orc_batch_mcts", i32, vp, P(TaflState), u32, u32, P(TaflMctsParams), u64, P(TaflRootChild), u32,
            P(u32), P(TaflMctsStats))
        _LIB = L
    return _LIB


def _rc_list(buf, n):
    return [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(n)]


def _copy_play(p: TaflPlay) -> TaflPlay:
    return TaflPlay(p.from_row, p.from_col, p.axis, p.disp)


class GameLogic:
    """GameLogic{rules, board_geo} — game/game/logic.rs:62-72."""

    def __init__(self, ruleset, side_len: int):
        L = lib()
        self.rules = ruleset
        self.side_len = side_len
        self._c_rules = ruleset.to_c() if isinstance(ruleset, abi.Ruleset) else ruleset
        self._buf = C.create_string_buffer(L.orc_sizeof_logic())
        if L.orc_logic_init(self._buf, C.byref(self._c_rules), side_len):
            raise ValueError("bad logic parameters")

    @property
    def ptr(self):
        return C.cast(self._buf, C.c_void_p)

    # -- movement ----------------------------------------------------------------------------
    def validate_play(self, play: TaflPlay, state: "GameState") -> int:
        """logic.rs:219-222 -> PlayInvalid code (0 = Ok(ValidPlay))."""
        return lib().orc_validate_play(self.ptr, state.ptr, play)

    def validate_play_for_side(self, play: TaflPlay, side: int, state: "GameState") -> int:
        return lib().orc_validate_play_for_side(self.ptr, state.ptr, play, side)

    def iter_plays(self, tile, state: "GameState"):
        """logic.rs:850-856; None = Err(BoardError::NoPiece)."""
        buf = (TaflPlay * 128)()
        n = lib().orc_iter_plays(self.ptr, state.ptr, tile[0], tile[1], buf, 128)
        if n < 0:
            return None
        return [_copy_play(buf[i]) for i in range(n)]

    def all_plays(self, state: "GameState"):
        """get_all_possible_moves — game/main.rs:33-43 (canonical order)."""
        buf = (TaflPlay * 1024)()
        n = lib().orc_all_plays(self.ptr, state.ptr, buf, 1024)
        return [_copy_play(buf[i]) for i in range(n)]

    def rollout_order_plays(self, state: "GameState"):
        buf = (TaflPlay * 1024)()
        n = lib().orc_rollout_order_plays(self.ptr, state.ptr, buf, 1024)
        return [_copy_play(buf[i]) for i in range(n)]

    def side_can_play(self, side: int, state: "GameState") -> bool:
        return bool(lib().orc_side_can_play(self.ptr, state.ptr, side))

    # -- captures / outcome ---------------------------------------------------------------------
    def get_captures(self, play: TaflPlay, moving_piece, state: "GameState"):
        """logic.rs:604-699; moving_piece = (piece_type, side); returns a set of (row, col)."""
        buf = (C.c_uint8 * 256)()
        n = lib().orc_get_captures(self.ptr, state.ptr, play, moving_piece[0], moving_piece[1], buf, 128)
        return set(_rc_list(buf, n))

    def detect_shieldwall(self, play: TaflPlay, state: "GameState"):
        """logic.rs:535-569; None or a set of tiles."""
        buf = (C.c_uint8 * 256)()
        n = lib().orc_detect_shieldwall(self.ptr, state.ptr, play, buf, 128)
        return None if n < 0 else set(_rc_list(buf, n))

    def find_enclosure(self, tile, enclosed: int, enclosing: int, abort_on_edge: bool, abort_on_corner: bool,
                       state: "GameState"):
        """logic.rs:309-401; None or an Enclosure."""
        e = Enclosure()
        ok = lib().orc_find_enclosure(self.ptr, state.ptr, tile[0], tile[1], enclosed, enclosing,
                                      int(abort_on_edge), int(abort_on_corner), e.ptr)
        return e if ok else None

    def enclosure_secure(self, encl: "Enclosure", inside_safe: bool, outside_safe: bool, state: "GameState") -> bool:
        return bool(lib().orc_enclosure_secure(self.ptr, state.ptr, encl.ptr, int(inside_safe), int(outside_safe)))

    def detect_exit_fort(self, state: "GameState") -> bool:
        return bool(lib().orc_detect_exit_fort(self.ptr, state.ptr))

    def do_play(self, play: TaflPlay, state: "GameState"):
        """logic.rs:827-834.  Returns (code, new_state, effects); the input state is not modified."""
        new = state.clone()
        eff = TaflEffects()
        code = lib().orc_do_play(self.ptr, new.ptr, play, C.byref(eff))
        return code, (new if code == 0 else state), eff

    def do_valid_play(self, play: TaflPlay, state: "GameState"):
        new = state.clone()
        eff = TaflEffects()
        lib().orc_do_valid_play(self.ptr, new.ptr, play, C.byref(eff))
        return new, eff

    def neighbors(self, tile):
        buf = (C.c_uint8 * 8)()
        n = lib().orc_neighbors(self.ptr, tile[0], tile[1], buf)
        return _rc_list(buf, n)

    def tiles_between(self, t1, t2):
        buf = (C.c_uint8 * 64)()
        n = lib().orc_tiles_between(self.ptr, t1[0], t1[1], t2[0], t2[1], buf)
        return _rc_list(buf, n)

    # -- build-defined rollout / MCTS ----------------------------------------------------------------
    def rollout(self, state: "GameState", seed: int, game_id: int, sim: int, max_plies: int) -> TaflRolloutResult:
        r = TaflRolloutResult()
        lib().orc_rollout(self.ptr, state.ptr, seed, game_id, sim, max_plies, C.byref(r))
        return r

    @staticmethod
    def state_hash(state: "GameState") -> int:
        """Leaf key of the search's playouts: predict(s) is the playout with simulation word sim_offset + state_hash(s)."""
        return int(lib().orc_state_hash(state.ptr))

    def random_advance(self, state: "GameState", seed: int, game_id: int, plies: int) -> "GameState":
        new = state.clone()
        lib().orc_random_advance(self.ptr, new.ptr, seed, game_id, plies)
        return new

    def gmcts(self, state: "GameState", n_sims: int, c_puct: float, predict, word_bits: int):
        """src/mcts.py:55-136 with an external predict(GameState) -> (priors float32[action_size], value).
        Returns (children [(TaflPlay, action, visits, q)], Ns[root], root priors float64 list, counts)."""
        L = lib()
        A = int(L.orc_action_size(self.side_len))
        size = int(L.orc_sizeof_state())

        def cb(_ctx, st_ptr, pri_out, val_out):
            g = GameState(None, word_bits=word_bits)
            C.memmove(g._buf, st_ptr, size)
            pri, v = predict(g)
            for a in range(A):
                pri_out[a] = float(pri[a])
            val_out[0] = float(v)

        fn = L._predict_t(cb)
        m = L.orc_gmcts_new(self.ptr, state.ptr, c_puct, fn, None)
        try:
            L.orc_gmcts_run(m, n_sims)
            buf = (TaflRootChild * 4096)()
            n = L.orc_gmcts_root_children(m, buf, 4096)
            kids = [(_copy_play(buf[i].play), int(buf[i].action), int(buf[i].visits), float(buf[i].q)) for i in range(n)]
            pri = (C.c_double * A)()
            L.orc_gmcts_root_priors(m, pri)
            cnt = (C.c_uint64 * 4)()
            L.orc_gmcts_counts(m, cnt)
            return kids, int(L.orc_gmcts_root_ns(m)), list(pri), list(cnt)
        finally:
            L.orc_gmcts_free(m)

    def mcts(self, state: "GameState", n_sims: int, c_puct: float, seed: int, max_rollout_plies: int,
             game_id: int = 0, sim_offset: int = 0):
        """Returns (children [(TaflPlay, action, visits, q)], Ns[root], stats)."""
        p = TaflMctsParams(n_sims, max_rollout_plies, c_puct, seed, sim_offset, 0)
        L = lib()
        m = L.orc_mcts_new(self.ptr, state.ptr, C.byref(p), game_id)
        try:
            L.orc_mcts_run(m)
            buf = (TaflRootChild * 1024)()
            n = L.orc_mcts_root_children(m, buf, 1024)
            kids = [(_copy_play(buf[i].play), int(buf[i].action), int(buf[i].visits), float(buf[i].q)) for i in range(n)]
            st = TaflMctsStats()
            L.orc_mcts_get_stats(m, C.byref(st))
            return kids, int(L.orc_mcts_root_ns(m)), st
        finally:
            L.orc_mcts_free(m)


class Enclosure:
    """logic.rs:24-38."""

    def __init__(self):
        self._buf = C.create_string_buffer(lib().orc_sizeof_enclosure())

    @property
    def ptr(self):
        return C.cast(self._buf, C.c_void_p)

    def _tiles(self, which):
        buf = (C.c_uint8 * (2 * 24 * 24))()
        n = lib().orc_enclosure_tiles(self.ptr, which, buf, 24 * 24)
        return set(_rc_list(buf, n))

    @property
    def occupied(self):
        return self._tiles(0)

    @property
    def unoccupied(self):
        return self._tiles(1)

    @property
    def boundary(self):
        return self._tiles(2)


class GameState:
    """GameState<T> — game/game/state.rs:119-146 over BitfieldBoardState<T> (game/board/state.rs:116-330)."""

    def __init__(self, fen: str | None, side_to_play: int = abi.ATTACKER, word_bits: int = 128):
        self.word_bits = word_bits
        self._buf = C.create_string_buffer(lib().orc_sizeof_state())
        if fen is not None:
            rc = lib().orc_state_init(self._buf, fen.encode(), side_to_play, word_bits)
            if rc:
                raise abi.ParseError({-1: "bad word size", -2: "BadChar", -3: "BadLineLen"}.get(rc, str(rc)))

    @property
    def ptr(self):
        return C.cast(self._buf, C.c_void_p)

    def clone(self) -> "GameState":
        g = GameState(None, word_bits=self.word_bits)
        C.memmove(g._buf, self._buf, len(self._buf))
        return g

    @classmethod
    def from_display_str(cls, s: str, word_bits: int = 64) -> "GameState":
        g = cls(None, word_bits=word_bits)
        rc = lib().orc_from_display_str(g._buf, s.encode(), word_bits)
        if rc:
            raise abi.ParseError(str(rc))
        return g

    @classmethod
    def from_abi(cls, st: TaflState, word_bits: int) -> "GameState":
        g = cls(None, word_bits=word_bits)
        if lib().orc_state_import(g._buf, C.byref(st), word_bits):
            raise ValueError("import failed")
        return g

    def to_abi(self) -> TaflState:
        st = TaflState()
        if lib().orc_state_export(self.ptr, C.byref(st)):
            raise ValueError("state wider than 256 bits cannot be exported")
        return st

    # board accessors (BoardState trait, game/board/state.rs:13-81)
    def get_piece(self, tile):
        v = lib().orc_get_piece(self.ptr, tile[0], tile[1])
        return None if v == 0 else (v & 0xFF, v >> 8)

    def set_piece(self, tile, piece):
        lib().orc_set_piece(self.ptr, tile[0], tile[1], piece[0], piece[1])

    def clear_tile(self, tile):
        lib().orc_clear_tile(self.ptr, tile[0], tile[1])

    def move_piece(self, frm, to):
        v = lib().orc_move_piece(self.ptr, frm[0], frm[1], to[0], to[1])
        if v < 0:
            raise RuntimeError("No piece to move.")
        return (v & 0xFF, v >> 8)

    def swap_pieces(self, t1, t2):
        lib().orc_swap_pieces(self.ptr, t1[0], t1[1], t2[0], t2[1])

    def get_king(self):
        v = lib().orc_get_king(self.ptr)
        return (v >> 8, v & 0xFF)

    def tile_occupied(self, tile) -> bool:
        return bool(lib().orc_tile_occupied(self.ptr, tile[0], tile[1]))

    def count_pieces(self, side: int) -> int:
        return lib().orc_count_pieces(self.ptr, side)

    def iter_occupied(self, side: int):
        buf = (C.c_uint8 * 1024)()
        n = lib().orc_iter_occupied(self.ptr, side, buf, 512)
        return _rc_list(buf, n)

    def board_to_matrix(self):
        """game/main.rs:55-83: n x n uint8 rows (corner 20, throne 30, soldier +1, king +5)."""
        n = self.to_abi().side_len if self.word_bits <= 256 else 0
        buf = (C.c_uint8 * (n * n))()
        lib().orc_board_to_matrix(self.ptr, buf)
        return [list(buf[r * n:(r + 1) * n]) for r in range(n)]

    def to_fen(self) -> str:
        buf = C.create_string_buffer(1024)
        lib().orc_to_fen(self.ptr, buf, 1024)
        return buf.value.decode()

    # plain fields through the ABI view (only <= 256-bit words)
    @property
    def side_to_play(self) -> int:
        return self.to_abi().side_to_play

    @side_to_play.setter
    def side_to_play(self, side: int):
        st = self.to_abi()
        st.side_to_play = side
        lib().orc_state_import(self._buf, C.byref(st), self.word_bits)

    @property
    def status(self):
        st = self.to_abi()
        return (st.status, st.reason, st.winner)


class RepetitionTracker:
    """game/game/state.rs:41-114."""

    def __init__(self):
        self._buf = C.create_string_buffer(lib().orc_sizeof_tracker())
        lib().orc_tracker_init(self._buf)

    def track_play(self, side: int, play: TaflPlay, captures: bool):
        lib().orc_tracker_track_play(self._buf, side, play, int(captures))

    def get_repetitions(self, side: int) -> int:
        return lib().orc_tracker_get_repetitions(self._buf, side)


class Game:
    """Game<T> — game/game/mod.rs:75-116 (history/undo included for the reference's test_undo)."""

    def __init__(self, ruleset, starting_board: str, word_bits: int | None = None):
        n = abi.fen_side_len(starting_board)
        wb = word_bits or abi.word_bits_for(n)
        self.state = GameState(starting_board, ruleset.starting_side, wb)
        self.logic = GameLogic(ruleset, n)
        self.play_history = []
        self.state_history = [self.state]

    def do_play(self, play: TaflPlay):
        code, new, eff = self.logic.do_play(play, self.state)
        if code != 0:
            return code, None
        self.state_history.append(self.state)
        self.state = new
        self.play_history.append((play, eff))
        return 0, new.status

    def undo_last_play(self):
        if self.state_history:
            self.state = self.state_history.pop()
            if self.play_history:
                self.play_history.pop()

    def iter_plays(self, tile):
        return self.logic.iter_plays(tile, self.state)


# ---- batch drivers (numpy-free; arrays are ctypes arrays of ABI structs) -----------------------------

def batch_movegen(logic: GameLogic, states, n: int, word_bits: int, want_masks: bool = True):
    counts = (C.c_uint32 * n)()
    mw = (abi.action_size(logic.side_len) + 31) // 32
    masks = (C.c_uint32 * (n * mw))() if want_masks else None
    rc = lib().orc_batch_movegen(logic.ptr, states, n, word_bits, counts, masks, mw)
    assert rc == 0
    return counts, masks


def batch_step(logic: GameLogic, states, n: int, word_bits: int, plays):
    eff = (TaflEffects * n)()
    assert lib().orc_batch_step(logic.ptr, states, n, word_bits, plays, eff) == 0
    return eff


def batch_step_kth(logic: GameLogic, states, n: int, word_bits: int, ranks):
    eff = (TaflEffects * n)()
    plays = (TaflPlay * n)()
    assert lib().orc_batch_step_kth(logic.ptr, states, n, word_bits, ranks, plays, eff) == 0
    return plays, eff


def batch_rollout(logic: GameLogic, states, n: int, word_bits: int, seed: int, sim: int, max_plies: int,
                  game_id_base: int = 0):
    out = (TaflRolloutResult * n)()
    assert lib().orc_batch_rollout(logic.ptr, states, n, word_bits, seed, sim, max_plies, game_id_base, out) == 0
    return out


def batch_random_advance(logic: GameLogic, states, n: int, word_bits: int, seed: int, plies, game_id_base: int = 0):
    assert lib().orc_batch_random_advance(logic.ptr, states, n, word_bits, seed, plies, game_id_base) == 0


def batch_mcts(logic: GameLogic, states, n: int, word_bits: int, params: TaflMctsParams, game_id_base: int = 0,
               max_children: int = 256):
    kids = (TaflRootChild * (n * max_children))()
    cnt = (C.c_uint32 * n)()
    stats = TaflMctsStats()
    assert lib().orc_batch_mcts(logic.ptr, states, n, word_bits, C.byref(params), game_id_base, kids,
                                max_children, cnt, C.byref(stats)) == 0
    return kids, cnt, stats


# ---- replay buffer text format: write_to_file (game/main.rs:86-132), statement by statement ------------------------
def write_to_file(file_path: str, matrix, vector, value1: int, value2: int, max_entries: int) -> None:
    import os as _os
    entries = []
    if _os.path.exists(file_path):                                   # main.rs:98
        with open(file_path, "r", newline="") as f:
            content = f.read()                                       # read_to_string, main.rs:99
        entries = _rust_lines(content)                               # content.lines(), main.rs:100
    if len(entries) >= max_entries:                                  # main.rs:104
        if entries:                                                  # Vec::remove(0) on an empty Vec would panic (max_entries == 0)
            entries.pop(0)                                           # main.rs:105: the oldest LINE
    new_entry = "{}\n{}\n{}\n{}".format(                             # main.rs:109-120
        "\n".join(",".join(str(int(v)) for v in row) for row in matrix),
        ",".join(str(int(v)) for v in vector), int(value1), int(value2))
    entries.append(new_entry)                                        # main.rs:122
    with open(file_path, "w", newline="") as f:                      # write + create + truncate, main.rs:125
        for e in entries:
            f.write(e + "\n")                                        # writeln!, main.rs:127-129


def _rust_lines(s: str):
    """str::lines(): split on '\\n', strip one trailing '\\r' per line, no empty element after a final newline."""
    if not s:
        return []
    parts = s.split("\n")
    if parts[-1] == "":
        parts.pop()
    return [p[:-1] if p.endswith("\r") else p for p in parts]
