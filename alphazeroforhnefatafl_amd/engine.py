"""Host-side mirror of the reference's `GameLogic` surface (game/game/logic.rs:62-880), batched.

`BatchedGameLogic(rules, side_len)` plays the role of `GameLogic::new(rules, board_length)`;
`GameBatch` holds n `GameState<T>` values (game/game/state.rs:119-146) resident in HBM.  Method names,
argument meaning and error behaviour follow the reference: per-game rule errors come back as
`PlayInvalid` codes (game/error.rs:49-70), never as exceptions; only library / HIP failures raise.
All compute runs in HIP kernels through the C-ABI (include/taflhip.h); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import weakref

from . import abi
from ._lib import check, lib
from .abi import (TaflEffects, TaflGmctsStats, TaflMctsParams, TaflMctsStats, TaflPlay, TaflRolloutResult, TaflRootChild, TaflState)

KC_MOVEGEN, KC_STEP, KC_ROLLOUT, KC_MCTS_TREE, KC_MCTS_ROLLOUT = range(5)


class BatchedGameLogic:
    """GameLogic{rules, board_geo} bound to one GPU (tafl_ctx)."""

    def __init__(self, rules: abi.Ruleset, side_len: int, word_bits: int | None = None, device: int = 0,
                 stream: int | None = None):
        self.rules = rules
        self.side_len = side_len
        self.word_bits = word_bits or abi.word_bits_for(side_len)
        self.device = device
        self._c_rules = rules.to_c()
        self._batches = weakref.WeakSet()   # the open GameBatch objects of this context (a batch leaves the set when it is closed or collected)
        self._h = C.c_void_p()
        check(lib().tafl_ctx_create(C.byref(self._c_rules), side_len, self.word_bits, device,
                                    C.c_void_p(stream) if stream else None, C.byref(self._h)))

    def close(self):
        """Destroys the context; batches created from it that are still open are closed first (the library refuses to destroy a
        context with live batches)."""
        if self._h:
            for b in list(self._batches):
                b.close()
            self._batches.clear()
            check(lib().tafl_ctx_destroy(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def action_size(self) -> int:
        return lib().tafl_action_size(self._h)

    @property
    def mask_words(self) -> int:
        return lib().tafl_action_mask_words(self._h)

    def sync(self):
        check(lib().tafl_sync(self._h))

    def new_batch(self, n_games: int, fen: str | None = None, side_to_play: int | None = None) -> "GameBatch":
        b = GameBatch(self, n_games)
        if fen is not None:
            b.reset_fen(fen, self.rules.starting_side if side_to_play is None else side_to_play)
        return b

    def state_from_fen(self, fen: str, side_to_play: int | None = None) -> TaflState:
        st = TaflState()
        side = self.rules.starting_side if side_to_play is None else side_to_play
        check(lib().tafl_state_from_fen(self._h, fen.encode(), side, C.byref(st)))
        return st

    # timing of the kernel classes (HIP events on the ctx stream)
    def timing_enable(self, on: bool = True):
        check(lib().tafl_timing_enable(self._h, int(on)))

    def timing_reset(self):
        check(lib().tafl_timing_reset(self._h))

    def timing_get(self, kernel_class: int):
        ms, k = C.c_double(), C.c_uint64()
        check(lib().tafl_timing_get(self._h, kernel_class, C.byref(ms), C.byref(k)))
        return ms.value, k.value

    def timing_get_union(self, kernel_class: int):
        """(time with at least one launch of the class in flight, sum of the launch durations) in ms since the last reset."""
        u, t = C.c_double(), C.c_double()
        check(lib().tafl_timing_get_union(self._h, kernel_class, C.byref(u), C.byref(t)))
        return u.value, t.value


class GameBatch:
    """n GameState<T> values in HBM + the batched GameLogic operations over them."""

    def __init__(self, logic: BatchedGameLogic, n_games: int):
        self.logic = logic
        self.n = n_games
        self._h = C.c_void_p()
        check(lib().tafl_batch_create(logic._h, n_games, C.byref(self._h)))
        logic._batches.add(self)

    def close(self):
        if self._h:
            lib().tafl_batch_destroy(self._h)
            self._h = C.c_void_p()
            self.logic._batches.discard(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- state movement -------------------------------------------------------------------------
    def reset_fen(self, fen: str, side_to_play: int):
        """GameState::new(fen, side) for every game (game/game/state.rs:136-145)."""
        check(lib().tafl_batch_reset_fen(self._h, fen.encode(), side_to_play))

    def upload(self, states, first: int = 0, count: int | None = None):
        count = self.n - first if count is None else count
        check(lib().tafl_batch_upload(self._h, states, first, count))

    def download(self, first: int = 0, count: int | None = None):
        count = self.n - first if count is None else count
        out = (TaflState * count)()
        check(lib().tafl_batch_download(self._h, out, first, count))
        return out

    # -- GameLogic surface -------------------------------------------------------------------------
    def iter_plays(self, want_masks: bool = True):
        """All legal plays of the side to move, per game: (counts, dense action masks) — get_all_possible_moves
        (game/main.rs:33-43) = iter_occupied x GameLogic::iter_plays (logic.rs:850-856)."""
        counts = (C.c_uint32 * self.n)()
        masks = (C.c_uint32 * (self.n * self.logic.mask_words))() if want_masks else None
        check(lib().tafl_movegen(self._h, counts, masks))
        return counts, masks

    def validate_play(self, plays):
        """GameLogic::validate_play (logic.rs:219-222): PlayInvalid code per game (0 = Ok)."""
        codes = (C.c_uint8 * self.n)()
        check(lib().tafl_validate(self._h, plays, codes))
        return codes

    def do_play(self, plays, want_effects: bool = True):
        """GameLogic::do_play (logic.rs:827-834) for every game; invalid plays leave that game unchanged."""
        eff = (TaflEffects * self.n)() if want_effects else None
        check(lib().tafl_step(self._h, plays, eff))
        return eff

    def do_kth_play(self, ranks):
        """Game i plays its (ranks[i] mod count)-th legal play in canonical order."""
        eff = (TaflEffects * self.n)()
        plays = (TaflPlay * self.n)()
        check(lib().tafl_step_kth(self._h, ranks, plays, eff))
        return plays, eff

    def side_can_play(self, side: int):
        """GameLogic::side_can_play (logic.rs:837-846)."""
        out = (C.c_uint8 * self.n)()
        check(lib().tafl_side_can_play(self._h, side, out))
        return out

    # -- rollouts / MCTS ------------------------------------------------------------------------------
    def rollout(self, seed: int, sim: int, max_plies: int, game_id_base: int = 0):
        out = (TaflRolloutResult * self.n)()
        check(lib().tafl_rollout(self._h, seed, sim, max_plies, game_id_base, out))
        return out

    def random_advance(self, seed: int, plies, game_id_base: int = 0):
        check(lib().tafl_random_advance(self._h, seed, plies, game_id_base))

    def mcts_reserve(self, max_sims: int):
        check(lib().tafl_mcts_reserve(self._h, max_sims))

    def mcts_run(self, n_sims: int, c_puct: float, seed: int, max_rollout_plies: int, game_id_base: int = 0,
                 sim_offset: int = 0, flags: int = 0):
        """`for i in range(numMCTSSims): self.search(canonicalBoard)` of MCTS.getActionProb (src/mcts.py:37-38) for every game.
        `flags`: abi.MCTS_FLAG_* semantics bits | abi.mcts_tune(pipeline, slots) execution tuning (never changes results)."""
        p = TaflMctsParams(n_sims, max_rollout_plies, c_puct, seed, sim_offset, flags)
        check(lib().tafl_mcts_run(self._h, C.byref(p), game_id_base))

    def mcts_run_async(self, n_sims: int, c_puct: float, seed: int, max_rollout_plies: int, game_id_base: int = 0,
                       sim_offset: int = 0, flags: int = 0, after: "GameBatch | None" = None):
        """The same search, enqueued on the batch's own streams without blocking the host (tafl_mcts_run_async): the search of another batch,
        a network, or the bookkeeping of the previous move run beside it.  `after`: hold this search back until that batch's search in
        flight is half-way through (two half-size batches started this way stay half a search apart, so that the nearly empty last rounds
        of one run under the full rounds of the other).  Join with mcts_wait(); every reader of the results joins by itself."""
        p = TaflMctsParams(n_sims, max_rollout_plies, c_puct, seed, sim_offset, flags)
        if after is None:
            check(lib().tafl_mcts_run_async(self._h, C.byref(p), game_id_base))
        else:
            check(lib().tafl_mcts_run_async_after(self._h, C.byref(p), game_id_base, after._h))

    def selfplay_run(self, n_moves: int, n_sims: int, c_puct: float, seed: int, max_rollout_plies: int, game_id_base: int = 0,
                     sim_offset: int = 0, flags: int = 0, want_plays: bool = True):
        """n_moves x (search + most visited play) per game on the device, every game at its own pace (tafl_selfplay_run): per game the same
        as `for m in range(n_moves): mcts_run(..., sim_offset=sim_offset + m * n_sims); mcts_play_best()`.  Returns the plays [m * n + g]."""
        p = TaflMctsParams(n_sims, max_rollout_plies, c_puct, seed, sim_offset, flags)
        plays = (TaflPlay * (self.n * n_moves))() if want_plays else None
        check(lib().tafl_selfplay_run(self._h, C.byref(p), n_moves, game_id_base, plays))
        return plays

    def mcts_wait(self):
        """Joins the search in flight; runs the rounds its slowest games still need (tafl_mcts_wait)."""
        check(lib().tafl_mcts_wait(self._h))

    def mcts_round_trace(self, cap: int = 4096):
        """(requested, run) playouts of every round of the last search (measurement; the two-kernel pipeline only)."""
        req, run, k = (C.c_uint32 * cap)(), (C.c_uint32 * cap)(), C.c_uint32()
        check(lib().tafl_mcts_round_trace(self._h, req, run, cap, C.byref(k)))
        m = min(k.value, cap)
        return list(req[:m]), list(run[:m])

    def mcts_stats(self) -> TaflMctsStats:
        st = TaflMctsStats()
        check(lib().tafl_mcts_get_stats(self._h, C.byref(st)))
        return st

    def mcts_root_children(self, max_children: int = 256):
        kids = (TaflRootChild * (self.n * max_children))()
        cnt = (C.c_uint32 * self.n)()
        check(lib().tafl_mcts_root_children(self._h, kids, max_children, cnt))
        return kids, cnt

    def mcts_root_visits(self):
        out = (C.c_uint32 * (self.n * self.logic.action_size))()
        check(lib().tafl_mcts_root_visits(self._h, out))
        return out

    def mcts_policy(self, temp: float = 1.0):
        out = (C.c_double * (self.n * self.logic.action_size))()
        check(lib().tafl_mcts_policy(self._h, temp, out))
        return out

    def encode_boards(self, out_device_ptr: int | None = None):
        """board_to_matrix (game/main.rs:55-83) for every game: uint8 [n, side_len, side_len].  With `out_device_ptr`
        (e.g. a torch uint8 tensor's data_ptr on this device) nothing crosses PCIe; otherwise returns a host ctypes array."""
        if out_device_ptr is not None:
            check(lib().tafl_encode_boards(self._h, C.c_void_p(out_device_ptr), 1))
            return None
        out = (C.c_uint8 * (self.n * self.logic.side_len * self.logic.side_len))()
        check(lib().tafl_encode_boards(self._h, C.cast(out, C.c_void_p), 0))
        return out

    def mcts_policy_device(self, temp: float = 1.0, out_device_ptr: int | None = None, tie_seed: int = 0, game_id_base: int = 0):
        """src/mcts.py:40-53 written by a kernel, any temp >= 0; float64 [n, action_size].  temp == 0: one-hot on the first maximum, or
        with tie_seed != 0 on the seeded choice among the maxima (the reference draws it with np.random.choice)."""
        if out_device_ptr is not None:
            check(lib().tafl_mcts_policy_device_ex(self._h, temp, tie_seed, game_id_base, C.c_void_p(out_device_ptr), 1))
            return None
        out = (C.c_double * (self.n * self.logic.action_size))()
        check(lib().tafl_mcts_policy_device_ex(self._h, temp, tie_seed, game_id_base, C.cast(out, C.c_void_p), 0))
        return out

    # -- guided MCTS: the caller's network is nnet.predict (src/mcts.py:85) --------------------------------------------
    def gmcts_begin(self, max_sims: int, edges_per_node: int = 256):
        check(lib().tafl_gmcts_begin(self._h, max_sims, edges_per_node))

    def gmcts_step(self, priors=None, values=None, c_puct: float = 1.0, n_sims: int = 64, device: bool = False, want_waiting: bool = True) -> int:
        """Expand the waiting leaves with (priors float32 [n, action_size], values float32 [n]) and select the next ones.
        `priors` / `values`: ctypes float arrays (host) or integer device pointers (device=True).  Returns the games now waiting."""
        w = C.c_uint32()
        pp = C.c_void_p(priors) if isinstance(priors, int) else (C.cast(priors, C.c_void_p) if priors is not None else None)
        pv = C.c_void_p(values) if isinstance(values, int) else (C.cast(values, C.c_void_p) if values is not None else None)
        check(lib().tafl_gmcts_step(self._h, pp, pv, int(device), c_puct, n_sims, C.byref(w) if want_waiting else None))
        return w.value

    def gmcts_leaves(self, boards_ptr: int | None = None, sides_ptr: int | None = None, waiting_ptr: int | None = None):
        """Network input of the waiting leaves: (boards uint8 [n, side, side], sides uint8 [n], waiting uint8 [n]); with device
        pointers nothing crosses PCIe, otherwise host ctypes arrays are returned."""
        if boards_ptr is not None:
            check(lib().tafl_gmcts_leaves(self._h, C.c_void_p(boards_ptr), C.c_void_p(sides_ptr), C.c_void_p(waiting_ptr), 1))
            return None
        n, s = self.n, self.logic.side_len
        boards, sides, waiting = (C.c_uint8 * (n * s * s))(), (C.c_uint8 * n)(), (C.c_uint8 * n)()
        check(lib().tafl_gmcts_leaves(self._h, C.cast(boards, C.c_void_p), C.cast(sides, C.c_void_p), C.cast(waiting, C.c_void_p), 0))
        return boards, sides, waiting

    def gmcts_root_children(self, max_children: int = 512):
        kids = (TaflRootChild * (self.n * max_children))()
        cnt = (C.c_uint32 * self.n)()
        check(lib().tafl_gmcts_root_children(self._h, kids, max_children, cnt))
        return kids, cnt

    def gmcts_root_visits(self, out_device_ptr: int | None = None):
        if out_device_ptr is not None:
            check(lib().tafl_gmcts_root_visits(self._h, C.c_void_p(out_device_ptr), 1))
            return None
        out = (C.c_uint32 * (self.n * self.logic.action_size))()
        check(lib().tafl_gmcts_root_visits(self._h, C.cast(out, C.c_void_p), 0))
        return out

    def gmcts_policy(self, temp: float = 1.0, out_device_ptr: int | None = None, tie_seed: int = 0, game_id_base: int = 0):
        if out_device_ptr is not None:
            check(lib().tafl_gmcts_policy_ex(self._h, temp, tie_seed, game_id_base, C.c_void_p(out_device_ptr), 1))
            return None
        out = (C.c_double * (self.n * self.logic.action_size))()
        check(lib().tafl_gmcts_policy_ex(self._h, temp, tie_seed, game_id_base, C.cast(out, C.c_void_p), 0))
        return out

    def gmcts_stats(self) -> TaflGmctsStats:
        st = TaflGmctsStats()
        check(lib().tafl_gmcts_get_stats(self._h, C.byref(st)))
        return st

    def mcts_play_best(self, want_results: bool = True):
        """Every game plays the most visited root play of its last search, on the device (tafl_mcts_play_best)."""
        if not want_results:
            check(lib().tafl_mcts_play_best(self._h, None, None))
            return None
        plays, eff = (TaflPlay * self.n)(), (TaflEffects * self.n)()
        check(lib().tafl_mcts_play_best(self._h, plays, eff))
        return plays, eff

    def mcts_best_play(self):
        plays = (TaflPlay * self.n)()
        visits = (C.c_uint32 * self.n)()
        check(lib().tafl_mcts_best_play(self._h, plays, visits))
        return plays, visits
