"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/taflhip.h declares, agrees on struct sizes, and FAILS LOUDLY without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from alphazeroforhnefatafl_amd import _lib, abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_library_builds_and_loads():
    _lib.build()
    L = _lib.lib()
    assert L.tafl_abi_version() == abi.ABI_VERSION


def test_every_declared_symbol_is_exported():
    hdr = open(os.path.join(ROOT, "include", "taflhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(tafl_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"tafl_status"}
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, (declared - bound, bound - declared)
    L = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name


def test_struct_sizes_match_header():
    # sizes are asserted at import of abi.py against the numbers documented in include/taflhip.h
    hdr = open(os.path.join(ROOT, "include", "taflhip.h")).read()
    for name, size in (("tafl_rules", 32), ("tafl_state", 104), ("tafl_effects", 40), ("tafl_rollout_result", 8),
                       ("tafl_root_child", 24)):
        assert re.search(r"\}\s*%s;\s*/\*\s*%d bytes" % (name, size), hdr), name
        assert abi.EXPECTED_SIZES[name] == size


def test_presets_from_library_match_host_mirror():
    L = _lib.lib()
    for name, r in abi.rules.BY_NAME.items():
        c = abi.TaflRules()
        assert L.tafl_preset_rules(name.encode(), C.byref(c)) == 0
        assert bytes(c) == bytes(r.to_c()), name
    for name in ("copenhagen", "brandubh", "magpie", "tablut", "copenhagen13"):
        assert L.tafl_preset_board(name.encode()).decode() == getattr(abi.boards, name.upper())
    assert L.tafl_preset_board(b"nope") is None
    assert L.tafl_preset_rules(b"nope", C.byref(abi.TaflRules())) != 0


@pytest.mark.skipif(_has_gpu(), reason="this check is for GPU-less boxes")
def test_no_cpu_fallback():
    from alphazeroforhnefatafl_amd.engine import BatchedGameLogic
    with pytest.raises(_lib.TaflError) as ei:
        BatchedGameLogic(abi.rules.COPENHAGEN, 11)
    assert ei.value.code == -2            # TAFL_ERR_NO_DEVICE
    assert "no CPU path" in str(ei.value)


def test_product_never_imports_oracle():
    """The oracle / host-sim are test infrastructure: nothing under the package may import, include, link or load them."""
    pkg = os.path.join(ROOT, "alphazeroforhnefatafl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith(".py"):
                txt = open(path).read()
                assert not re.search(r"^\s*(from|import)\s+(oracle|tests)\b", txt, flags=re.M), path
            elif f.endswith((".hip", ".hpp", ".h", ".cpp")):
                txt = open(path).read()
                assert not re.search(r"#\s*include\s+[\"<][^\">]*(oracle|hostsim)", txt), path
            elif f == "Makefile":
                txt = open(path).read()
                assert "oracle" not in txt and "hostsim" not in txt, path
            else:
                continue
            assert "liboracle" not in txt and "libhostsim" not in txt, path


def test_state_to_fen_round_trips_against_the_oracle():
    """tafl_state_to_fen (BoardState::to_fen, board/state.rs:271-295) vs the oracle's to_fen on start boards and random positions."""
    import random
    from alphazeroforhnefatafl_amd import abi
    from oracle import oracle as orc
    from tests import parity_util as pu
    rng = random.Random(3)
    for name, (rules, fen, wb) in pu.CONFIGS.items():
        n = abi.fen_side_len(fen)
        st = orc.GameState(fen, rules.starting_side, wb)
        assert abi.state_to_fen(st.to_abi(), wb) == fen == st.to_fen(), name
        states = pu.random_board_states(rng, n, wb, 40)
        for g in range(40):
            assert abi.state_to_fen(states[g], wb) == orc.GameState.from_abi(states[g], wb).to_fen(), (name, g)
