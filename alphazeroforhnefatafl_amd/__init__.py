"""alphazeroforhnefatafl_amd — MI355X-native batched Hnefatafl environment + MCTS rollout engine.

One hot path of payelmuk91/AlphaZeroForHnefatafl (move generation, env step, random rollout, MCTS over tens of
thousands of concurrent games) as hand-written HIP kernels behind a C-ABI (include/taflhip.h).  See DESIGN.md.
"""
from . import abi
from .abi import Ruleset, boards, rules

__all__ = ["abi", "Ruleset", "boards", "rules", "BatchedGameLogic", "GameBatch", "MCTS", "MCTSArgs", "GuidedMCTS", "BatchedGame"]


def __getattr__(name):
    # engine/mcts load libtaflhip.so (fails loudly if it is missing); abi is importable without it
    if name in ("BatchedGameLogic", "GameBatch"):
        from . import engine
        return getattr(engine, name)
    if name in ("MCTS", "MCTSArgs", "GuidedMCTS"):
        from . import mcts
        return getattr(mcts, name)
    if name == "BatchedGame":
        from . import game
        return game.BatchedGame
    raise AttributeError(name)
