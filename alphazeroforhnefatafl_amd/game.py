"""Batched mirror of the reference's `Game<T>` (game/game/mod.rs:75-116): rules + current states + play / state history
with undo, for n games at once.  Host-side bookkeeping around `GameBatch` (the states themselves live in HBM and every
rule evaluation runs in the HIP kernels); not part of the hot path.

Per game the semantics are the reference's: `Game::new` starts `state_history` with the initial state (mod.rs:91), a
successful `do_play` pushes the state it replaces and the play record (mod.rs:97-102), a rejected play changes nothing,
`undo_last_play` pops one state and one record (mod.rs:104-109)."""
from __future__ import annotations

import ctypes as C

from . import abi
from .abi import TaflEffects, TaflPlay, TaflState
from .engine import BatchedGameLogic, GameBatch


class BatchedGame:
    def __init__(self, ruleset: abi.Ruleset, starting_board: str, n_games: int, word_bits: int | None = None, device: int = 0):
        n = abi.fen_side_len(starting_board)
        self.logic = BatchedGameLogic(ruleset, n, word_bits, device)
        self.batch: GameBatch = self.logic.new_batch(n_games, starting_board)
        self.n = n_games
        self._snaps = [self.batch.download()]                      # snapshot 0 = Game::new's state
        self._stack = [[0] for _ in range(n_games)]                # state_history per game, as snapshot ids
        self.play_history = [[] for _ in range(n_games)]           # (TaflPlay, TaflEffects) per game

    @property
    def state(self):
        """Current GameState of every game (ABI structs)."""
        return self.batch.download()

    def do_play(self, plays):
        """Game::do_play for every game: returns (codes, effects); codes[g] = PlayInvalid (0 = Ok, status in effects[g].status)."""
        snap = self.batch.download()
        eff = self.batch.do_play(plays)
        k = None
        for g in range(self.n):
            if eff[g].code == 0:
                if k is None:
                    self._snaps.append(snap)
                    k = len(self._snaps) - 1
                self._stack[g].append(k)
                p, e = TaflPlay(), TaflEffects()
                C.memmove(C.byref(p), C.byref(plays[g]), C.sizeof(TaflPlay))
                C.memmove(C.byref(e), C.byref(eff[g]), C.sizeof(TaflEffects))
                self.play_history[g].append((p, e))
        return [eff[g].code for g in range(self.n)], eff

    def undo_last_play(self, games=None):
        """Game::undo_last_play for the listed games (default: all)."""
        cur = self.batch.download()
        changed = False
        for g in (range(self.n) if games is None else games):
            if self._stack[g]:
                k = self._stack[g].pop()
                C.memmove(C.byref(cur[g]), C.byref(self._snaps[k][g]), C.sizeof(TaflState))
                if self.play_history[g]:
                    self.play_history[g].pop()
                changed = True
        if changed:
            self.batch.upload(cur)

    def iter_plays(self):
        """Legal plays of the side to move, per game: (counts, dense action masks) - GameBatch.iter_plays."""
        return self.batch.iter_plays()
