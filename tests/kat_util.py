"""Helpers to read tests/golden/reference_kats.json (shared by the oracle and GPU replays)."""
import json
import os

from alphazeroforhnefatafl_amd import abi

HERE = os.path.dirname(os.path.abspath(__file__))

with open(os.path.join(HERE, "golden", "reference_kats.json")) as f:
    KATS = json.load(f)

PT = {"King": abi.KING, "Soldier": abi.SOLDIER, "Knight": abi.KNIGHT, "Commander": abi.COMMANDER,
      "Guard": abi.GUARD, "Mercenary": abi.MERCENARY}
SIDE = {"A": abi.ATTACKER, "D": abi.DEFENDER}
THRONE = {"NoThrone": abi.NO_THRONE, "NoPass": abi.NO_PASS, "KingPass": abi.KING_PASS, "NoEntry": abi.NO_ENTRY,
          "KingEntry": abi.KING_ENTRY}
REASON_CODE = {n: i for i, n in enumerate(abi.PLAY_INVALID_NAMES)}


def pieceset(spec: str) -> int:
    if spec == "none":
        return abi.ps_none()
    if spec == "all":
        return abi.ps_all()
    kind, _, rest = spec.partition(":")
    if kind == "type":
        return abi.ps_type(PT[rest])
    if kind == "piece":
        t, s = rest.split(":")
        return abi.ps_piece(PT[t], SIDE[s])
    if kind == "pieces":
        v = 0
        for item in rest.split(","):
            t, s = item.split(":")
            v |= abi.ps_piece(PT[t], SIDE[s])
        return v
    if kind == "side":
        return abi.ps_side(SIDE[rest])
    raise ValueError(spec)


def ruleset(name: str) -> abi.Ruleset:
    if name in abi.rules.BY_NAME:
        return abi.rules.BY_NAME[name]
    var = KATS["rule_variants"][name]
    r = abi.rules.BY_NAME[var["base"]]
    kw = {}
    for k, v in var["override"].items():
        if k == "throne_movement":
            kw[k] = THRONE[v]
        elif k == "shieldwall":
            kw[k] = (bool(v[0]), pieceset(v[1]))
        else:
            kw[k] = pieceset(v)
    return r.replace(**kw)


def side_of(spec: str, rules: abi.Ruleset) -> int:
    return rules.starting_side if spec == "starting" else SIDE[spec]


def play(p) -> abi.TaflPlay:
    return abi.play_from_tiles((p[0], p[1]), (p[2], p[3]))


def piece(ch: str):
    return {"t": (abi.SOLDIER, abi.ATTACKER), "T": (abi.SOLDIER, abi.DEFENDER), "K": (abi.KING, abi.DEFENDER)}[ch]


def status_tuple(spec):
    """["Win","KingCaptured","A"] -> (status, reason, winner)."""
    if spec[0] == "Ongoing":
        return (abi.ONGOING, 0, 0)
    if spec[0] == "Win":
        return (abi.WIN, abi.WIN_REASON_NAMES.index(spec[1]), SIDE[spec[2]])
    return (abi.DRAW, abi.DRAW_REASON_NAMES.index(spec[1]), 0)


def tiles(lst):
    return set(tuple(t) for t in lst)
