"""Batched mirror of the reference's `MCTS` class (src/mcts.py:11-136).

`MCTS(game_batch, args).getActionProb(temp)` runs `args.numMCTSSims` simulations for every game of
the batch on the GPU (select / expand / random-rollout / backup as lock-step HIP kernels) and returns the
policy vectors of src/mcts.py:40-53.  Without a network, `predict` is the random-rollout mode of SURVEY.md §8a (uniform priors over legal plays + the value of
one seeded random playout).  With `nnet`, `predict` is the caller's network evaluated for the whole batch at once
(`GuidedMCTS`, include/taflhip.h "guided MCTS"): nnet.predict_batch(boards, sides, waiting) -> (priors, values).
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


class GuidedMCTS:
    """src/mcts.py:11-136 with `nnet` as the evaluator, for every game of the batch in lock step.

    `nnet.predict_batch(boards, sides, waiting)` receives the network input of the waiting leaves — the
    board_to_matrix planes uint8 [n, side, side] (game/main.rs:55-83), the side to move [n] and a waiting flag [n] — and
    returns (priors float32 [n, action_size], values float32 [n]); rows of games that are not waiting are ignored.
    With `device=True` the three inputs are integer device pointers into buffers the caller allocated
    (`buffers=(boards_ptr, sides_ptr, waiting_ptr)`) and the outputs are device pointers too: nothing crosses PCIe.
    """

    def __init__(self, batch: GameBatch, nnet, args: MCTSArgs, edges_per_node: int = 256, device: bool = False, buffers=None):
        self.batch, self.nnet, self.args = batch, nnet, args
        self.edges_per_node, self.device, self.buffers = edges_per_node, device, buffers
        self.rounds = 0
        self._ran = False

    def search_all(self):
        a, b = self.args, self.batch
        b.gmcts_begin(a.numMCTSSims, self.edges_per_node)
        waiting = b.gmcts_step(None, None, a.cpuct, a.numMCTSSims)
        while waiting:
            if self.device:
                b.gmcts_leaves(*self.buffers)
                priors, values = self.nnet.predict_batch(*self.buffers)
            else:
                priors, values = self.nnet.predict_batch(*b.gmcts_leaves())
            waiting = b.gmcts_step(priors, values, a.cpuct, a.numMCTSSims, device=self.device)
            self.rounds += 1
        self._ran = True

    def getActionProb(self, temp: float = 1.0):
        """src/mcts.py:28-53 per game: flat float64 [n_games * action_size] (temp 0: first maximum)."""
        self.search_all()
        return self.batch.gmcts_policy(temp)

    def root_children(self, max_children: int = 512):
        if not self._ran:
            self.search_all()
        return self.batch.gmcts_root_children(max_children)
