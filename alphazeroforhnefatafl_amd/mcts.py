"""Batched mirror of the reference's `MCTS` class (src/mcts.py:11-136).

`MCTS(game_batch, args).getActionProb(temp)` runs `args.numMCTSSims` simulations for every game of
the batch on the GPU (select / expand / random-rollout / backup as lock-step HIP kernels) and returns the
policy vectors of src/mcts.py:40-53.  `predict` is fixed to the random-rollout mode of SURVEY.md §8a:
uniform priors over legal plays + the value of one seeded random playout.
"""
from __future__ import annotations

from dataclasses import dataclass

from .engine import GameBatch


@dataclass
class MCTSArgs:
    numMCTSSims: int = 64      # src/mcts.py:37
    cpuct: float = 1.0         # src/mcts.py:112
    seed: int = 0
    max_rollout_plies: int = 512
    game_id_base: int = 0


class MCTS:
    def __init__(self, batch: GameBatch, args: MCTSArgs):
        self.batch = batch
        self.args = args
        self._ran = False

    def search_all(self):
        """`for i in range(numMCTSSims): self.search(canonicalBoard)` (src/mcts.py:37-38) for every game."""
        a = self.args
        self.batch.mcts_run(a.numMCTSSims, a.cpuct, a.seed, a.max_rollout_plies, a.game_id_base)
        self._ran = True

    def getActionProb(self, temp: float = 1.0):
        """src/mcts.py:28-53.  Returns a flat ctypes array [n_games * action_size] of float64.
        temp == 0 puts all mass on the FIRST maximum (the reference draws among ties with np.random)."""
        self.search_all()
        return self.batch.mcts_policy(temp)

    def root_children(self, max_children: int = 256):
        if not self._ran:
            self.search_all()
        return self.batch.mcts_root_children(max_children)

    def best_play(self):
        """src/mcts.rs:216-227 best_move (max visits)."""
        if not self._ran:
            self.search_all()
        return self.batch.mcts_best_play()
