"""ctypes front-end of tests/hostsim/libhostsim.so (TEST HARNESS: product device code compiled for the host)."""
import ctypes as C
import os
import subprocess

from alphazeroforhnefatafl_amd import abi
from alphazeroforhnefatafl_amd.abi import (TaflEffects, TaflMctsParams, TaflMctsStats, TaflPlay, TaflRolloutResult,
                                          TaflRootChild, TaflRules, TaflState)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        subprocess.check_call(["make", "-C", _HERE, "-s", "libhostsim.so"])
        L = C.CDLL(os.path.join(_HERE, "libhostsim.so"))
        P, u8, u32, u64, i32 = C.POINTER, C.c_uint8, C.c_uint32, C.c_uint64, C.c_int
        head = [P(TaflRules), u8, u32]

        def sig(name, *args):
            f = getattr(L, name)
            f.restype = i32
            f.argtypes = head + list(args)

        sig("hs_movegen", P(TaflState), u32, P(u32), P(u32), u32)
        sig("hs_validate", P(TaflState), u32, P(TaflPlay), P(u8))
        sig("hs_step", P(TaflState), u32, P(TaflPlay), P(TaflEffects))
        sig("hs_step_kth", P(TaflState), u32, P(u32), P(TaflPlay), P(TaflEffects))
        sig("hs_side_can_play", P(TaflState), u32, u8, P(u8))
        sig("hs_rollout", P(TaflState), u32, u64, u32, u32, u64, P(TaflRolloutResult))
        sig("hs_random_advance", P(TaflState), u32, u64, P(u32), u64)
        sig("hs_mcts", P(TaflState), u32, P(TaflMctsParams), u64, P(TaflRootChild), u32, P(u32), P(TaflMctsStats))
        vp = C.c_void_p
        L.hs_gmcts_new.restype = vp
        L.hs_gmcts_new.argtypes = head + [P(TaflState), u32, u32, u32]
        L.hs_gmcts_free.restype = None
        L.hs_gmcts_free.argtypes = [vp]
        L.hs_gmcts_step.restype = u32
        L.hs_gmcts_step.argtypes = [vp, P(C.c_float), P(C.c_float), C.c_double, u32]
        L.hs_gmcts_leaves.restype = None
        L.hs_gmcts_leaves.argtypes = [vp, P(u8), P(u8), P(u8)]
        L.hs_gmcts_root_children.restype = None
        L.hs_gmcts_root_children.argtypes = [vp, P(TaflRootChild), u32, P(u32)]
        L.hs_gmcts_counts.restype = None
        L.hs_gmcts_counts.argtypes = [vp, P(u64)]
        L.hs_set_scenarios.restype = None
        L.hs_set_scenarios.argtypes = [C.c_uint32]
        L.hs_set_spec_target.restype = None
        L.hs_set_spec_target.argtypes = [C.c_uint32]
        L.hs_set_dense13.restype = None
        L.hs_set_dense13.argtypes = [C.c_int]
        L.hs_set_log_cap.restype = None
        L.hs_set_log_cap.argtypes = [C.c_uint32]
        L.hs_set_capacity.restype = None
        L.hs_set_capacity.argtypes = [C.c_uint32]
        L.hs_round_work.restype = C.c_uint32
        L.hs_round_work.argtypes = [C.POINTER(C.c_uint32), C.c_uint32]
        L.hs_selfplay.restype = C.c_int
        L.hs_selfplay.argtypes = [C.POINTER(TaflRules), C.c_uint8, C.c_uint32, C.POINTER(TaflState), C.c_uint32,
                                  C.POINTER(TaflMctsParams), C.c_uint64, C.c_uint32, C.POINTER(TaflPlay), C.POINTER(TaflMctsStats)]
        L.hs_set_spec_k.restype = None
        L.hs_set_spec_k.argtypes = [C.c_uint32]
        L.hs_force_generic.restype = None
        L.hs_force_generic.argtypes = [C.c_int]
        _LIB = L
    return _LIB


class HostSim:
    def __init__(self, ruleset, side_len, word_bits):
        self.rules = ruleset.to_c() if isinstance(ruleset, abi.Ruleset) else ruleset
        self.n = side_len
        self.wb = word_bits
        self.mw = (abi.action_size(side_len) + 31) // 32

    def _h(self):
        return (C.byref(self.rules), self.n, self.wb)

    def movegen(self, states, n, want_masks=True):
        counts = (C.c_uint32 * n)()
        masks = (C.c_uint32 * (n * self.mw))() if want_masks else None
        assert lib().hs_movegen(*self._h(), states, n, counts, masks, self.mw) == 0
        return counts, masks

    def validate(self, states, n, plays):
        codes = (C.c_uint8 * n)()
        assert lib().hs_validate(*self._h(), states, n, plays, codes) == 0
        return codes

    def step(self, states, n, plays):
        eff = (TaflEffects * n)()
        assert lib().hs_step(*self._h(), states, n, plays, eff) == 0
        return eff

    def step_kth(self, states, n, ranks):
        eff = (TaflEffects * n)()
        plays = (TaflPlay * n)()
        assert lib().hs_step_kth(*self._h(), states, n, ranks, plays, eff) == 0
        return plays, eff

    def side_can_play(self, states, n, side):
        out = (C.c_uint8 * n)()
        assert lib().hs_side_can_play(*self._h(), states, n, side, out) == 0
        return out

    def rollout(self, states, n, seed, sim, max_plies, base=0):
        out = (TaflRolloutResult * n)()
        assert lib().hs_rollout(*self._h(), states, n, seed, sim, max_plies, base, out) == 0
        return out

    def random_advance(self, states, n, seed, plies, base=0):
        assert lib().hs_random_advance(*self._h(), states, n, seed, plies, base) == 0

    def selfplay(self, states, n, params, n_moves, base=0):
        """tafl_selfplay_run on the host: `states` is advanced in place; returns (plays [n_moves * n], stats over all searches)."""
        plays = (TaflPlay * (n * n_moves))()
        stats = TaflMctsStats()
        assert lib().hs_selfplay(*self._h(), states, n, C.byref(params), base, n_moves, plays, C.byref(stats)) == 0
        return plays, stats

    def mcts(self, states, n, params, base=0, max_children=256):
        kids = (TaflRootChild * (n * max_children))()
        cnt = (C.c_uint32 * n)()
        stats = TaflMctsStats()
        assert lib().hs_mcts(*self._h(), states, n, C.byref(params), base, kids, max_children, cnt, C.byref(stats)) == 0
        return kids, cnt, stats


def force_generic(on: bool):
    """Differential tests: make host-sim playouts use the generic Engine::rollout instead of the fast engine."""
    lib().hs_force_generic(int(on))


def set_log_cap(cap: int):
    """Undo records of each kind a prediction pass may write in host-sim MCTS runs (the device: 16 per lane in LDS, 5 in the fused kernel);
    an overflow ends the pass early - fewer predictions, same results."""
    lib().hs_set_log_cap(cap)


def set_spec_k(k: int, target: int = 0, capacity: int = 0):
    """Playout slots per game that exist in host-sim MCTS runs (1 = no speculation), the slots per game and round the
    search is planned for (0 = no plan: every game issues what its own hit history allows), and the playouts one round may
    run (0 = all requested; otherwise the rest waits for the next round, as on a full device)."""
    lib().hs_set_spec_k(k)
    lib().hs_set_spec_target(target)
    lib().hs_set_capacity(capacity)


def set_scenarios(s: int):
    """Scenario passes of every prediction (1 or 2); 0 = the product's policy (Ops::mcts_scenarios)."""
    lib().hs_set_scenarios(s)


def round_work():
    """Playouts executed in each round of the last HostSim.mcts call."""
    buf = (C.c_uint32 * 4096)()
    n = lib().hs_round_work(buf, 4096)
    return list(buf[:min(n, 4096)])


def set_dense13(on: bool):
    """Rollouts / searches of 13x13 positions in the dense 13-column layout (what the library does for the 13x13 preset)."""
    lib().hs_set_dense13(int(on))
