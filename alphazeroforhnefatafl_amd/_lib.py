"""Loader of libtaflhip.so (the HIP kernels + C-ABI).  Fails loudly: there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

from . import abi
from .abi import (TaflEffects, TaflGmctsStats, TaflMctsParams, TaflMctsStats, TaflPlay, TaflRolloutResult, TaflRootChild, TaflRules,
                  TaflState)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtaflhip.so")      # the in-tree build; measurement scripts that A/B another build assign LIB_PATH before lib()
_LIB = None

# every symbol include/taflhip.h declares: (name, restype, argtypes)
_vp, _u8, _u32, _u64, _i32, _dbl = C.c_void_p, C.c_uint8, C.c_uint32, C.c_uint64, C.c_int, C.c_double
_P = C.POINTER
SYMBOLS = [
    ("tafl_ctx_create", _i32, [_P(TaflRules), _u8, _u32, _i32, _vp, _P(_vp)]),
    ("tafl_ctx_destroy", _i32, [_vp]),
    ("tafl_last_error", C.c_char_p, []),
    ("tafl_abi_version", _i32, []),
    ("tafl_preset_rules", _i32, [C.c_char_p, _P(TaflRules)]),
    ("tafl_preset_board", C.c_char_p, [C.c_char_p]),
    ("tafl_action_size", _u32, [_vp]),
    ("tafl_action_mask_words", _u32, [_vp]),
    ("tafl_action_encode", _i32, [_vp, TaflPlay, _P(_u32)]),
    ("tafl_action_decode", _i32, [_vp, _u32, _P(TaflPlay)]),
    ("tafl_batch_create", _i32, [_vp, _u32, _P(_vp)]),
    ("tafl_batch_destroy", _i32, [_vp]),
    ("tafl_batch_size", _u32, [_vp]),
    ("tafl_batch_reset_fen", _i32, [_vp, C.c_char_p, _u8]),
    ("tafl_batch_upload", _i32, [_vp, _P(TaflState), _u32, _u32]),
    ("tafl_batch_download", _i32, [_vp, _P(TaflState), _u32, _u32]),
    ("tafl_state_from_fen", _i32, [_vp, C.c_char_p, _u8, _P(TaflState)]),
    ("tafl_state_to_fen", _i32, [_P(TaflState), _u32, C.c_char_p, _u32]),
    ("tafl_sync", _i32, [_vp]),
    ("tafl_movegen", _i32, [_vp, _P(_u32), _P(_u32)]),
    ("tafl_validate", _i32, [_vp, _P(TaflPlay), _P(_u8)]),
    ("tafl_step", _i32, [_vp, _P(TaflPlay), _P(TaflEffects)]),
    ("tafl_step_kth", _i32, [_vp, _P(_u32), _P(TaflPlay), _P(TaflEffects)]),
    ("tafl_side_can_play", _i32, [_vp, _u8, _P(_u8)]),
    ("tafl_rollout", _i32, [_vp, _u64, _u32, _u32, _u64, _P(TaflRolloutResult)]),
    ("tafl_random_advance", _i32, [_vp, _u64, _P(_u32), _u64]),
    ("tafl_mcts_reserve", _i32, [_vp, _u32]),
    ("tafl_mcts_run", _i32, [_vp, _P(TaflMctsParams), _u64]),
    ("tafl_mcts_run_async", _i32, [_vp, _P(TaflMctsParams), _u64]),
    ("tafl_mcts_run_async_after", _i32, [_vp, _P(TaflMctsParams), _u64, _vp]),
    ("tafl_mcts_wait", _i32, [_vp]),
    ("tafl_selfplay_run", _i32, [_vp, _P(TaflMctsParams), _u32, _u64, _P(TaflPlay)]),
    ("tafl_mcts_get_stats", _i32, [_vp, _P(TaflMctsStats)]),
    ("tafl_mcts_root_children", _i32, [_vp, _P(TaflRootChild), _u32, _P(_u32)]),
    ("tafl_mcts_root_visits", _i32, [_vp, _P(_u32)]),
    ("tafl_mcts_policy", _i32, [_vp, _dbl, _P(_dbl)]),
    ("tafl_mcts_best_play", _i32, [_vp, _P(TaflPlay), _P(_u32)]),
    ("tafl_mcts_play_best", _i32, [_vp, _P(TaflPlay), _P(TaflEffects)]),
    ("tafl_encode_boards", _i32, [_vp, _vp, _i32]),
    ("tafl_mcts_policy_device", _i32, [_vp, _dbl, _vp, _i32]),
    ("tafl_mcts_policy_device_ex", _i32, [_vp, _dbl, _u64, _u64, _vp, _i32]),
    ("tafl_gmcts_begin", _i32, [_vp, _u32, _u32]),
    ("tafl_gmcts_step", _i32, [_vp, _vp, _vp, _i32, _dbl, _u32, _P(_u32)]),
    ("tafl_gmcts_leaves", _i32, [_vp, _vp, _vp, _vp, _i32]),
    ("tafl_gmcts_root_children", _i32, [_vp, _P(TaflRootChild), _u32, _P(_u32)]),
    ("tafl_gmcts_root_visits", _i32, [_vp, _vp, _i32]),
    ("tafl_gmcts_policy", _i32, [_vp, _dbl, _vp, _i32]),
    ("tafl_gmcts_policy_ex", _i32, [_vp, _dbl, _u64, _u64, _vp, _i32]),
    ("tafl_gmcts_get_stats", _i32, [_vp, _P(TaflGmctsStats)]),
    ("tafl_replay_append", _i32, [C.c_char_p, _P(_u8), _u8, _P(_u8), _u32, _u8, _u8, _u64]),
    ("tafl_replay_append_batch", _i32, [C.c_char_p, _P(_u8), _u8, _u32, _P(_u8), _P(_u32), _P(_u8), _P(_u8), _u64]),
    ("tafl_replay_read", _i32, [C.c_char_p, _u8, _u32, _P(_u8), _P(_u8), _u32, _P(_u32), _P(_u8), _P(_u8), _P(_u32)]),
    ("tafl_mcts_round_trace", _i32, [_vp, _P(_u32), _P(_u32), _u32, _P(_u32)]),
    ("tafl_timing_enable", _i32, [_vp, _i32]),
    ("tafl_timing_reset", _i32, [_vp]),
    ("tafl_timing_get", _i32, [_vp, _i32, _P(_dbl), _P(_u64)]),
    ("tafl_timing_get_union", _i32, [_vp, _i32, _P(_dbl), _P(_dbl)]),
    ("tafl_ctx_stream", _vp, [_vp]),
]


class TaflError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"taflhip error {code}: {msg}")
        self.code = code


def build(force: bool = False) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.check_call(["make", "-C", csrc, "-s"])
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              f"(make -C alphazeroforhnefatafl_amd/csrc). There is no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64, and if this library pulled
        # in /opt/rocm's copies first a later `import torch` would find no GPU.  Loading torch first (when it is installed)
        # makes both share torch's runtime.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            f = getattr(L, name)          # AttributeError if the symbol is not exported
            f.restype = res
            f.argtypes = args
        if L.tafl_abi_version() != abi.ABI_VERSION:
            raise ImportError("libtaflhip.so ABI version mismatch")
        _LIB = L
    return _LIB


def check(rc: int):
    if rc != 0:
        raise TaflError(rc, lib().tafl_last_error().decode(errors="replace"))
