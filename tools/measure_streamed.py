#!/usr/bin/env python3
"""Kernel-only rates of the streamed entry points (BASELINE configs[1]: move-gen + step over a batch of 11x11 Copenhagen
games advanced by (i mod 64) seeded random plies), measured with the library's HIP-event timers.  Not the bench line
(bench.py is BASELINE's headline metric); numbers are quoted in DESIGN.md section 6."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazeroforhnefatafl_amd import abi  # noqa: E402
from alphazeroforhnefatafl_amd.engine import KC_MOVEGEN, KC_ROLLOUT, KC_STEP, BatchedGameLogic  # noqa: E402


def main():
    out = []
    for name, rules, fen, n, wb in (("copenhagen11", abi.rules.COPENHAGEN, abi.boards.COPENHAGEN, 11, 128),
                                    ("brandubh7", abi.rules.BRANDUBH, abi.boards.BRANDUBH, 7, 64),
                                    ("copenhagen13", abi.rules.COPENHAGEN, abi.boards.COPENHAGEN13, 13, 256)):
        logic = BatchedGameLogic(rules, n, wb)
        for G in (4096, 65536):
            b = logic.new_batch(G, fen)
            plies = (C.c_uint32 * G)(*[i % 64 for i in range(G)])
            b.random_advance(1, plies, 0)
            ranks = (C.c_uint32 * G)(*[(i * 2654435761) & 0x3FFFFFFF for i in range(G)])
            b.iter_plays(want_masks=False); b.iter_plays(want_masks=True)      # warm-up
            logic.timing_reset(); logic.timing_enable(True)
            for _ in range(5):
                b.iter_plays(want_masks=False)
            ms_c, k_c = logic.timing_get(KC_MOVEGEN)
            logic.timing_reset()
            for _ in range(5):
                b.iter_plays(want_masks=True)
            ms_m, k_m = logic.timing_get(KC_MOVEGEN)
            # env step with GIVEN plays (BASELINE config 2 ii): the plays are each game's (r mod count)-th legal play, obtained
            # once on a scratch copy; the timed call is tafl_step = validate + do_valid_play on the original states
            states = b.download()
            scratch = logic.new_batch(G)
            scratch.upload(states)
            logic.timing_reset()
            plays, _ = scratch.do_kth_play(ranks)
            ms_k, k_k = logic.timing_get(KC_STEP)                # tafl_step_kth: legal mask in LDS, k-th set bit, do_valid_play
            scratch.close()
            logic.timing_reset()
            ms_s = k_s = 0
            for _ in range(3):
                b.upload(states)
                logic.timing_reset()
                b.do_play(plays)
                m1, k1 = logic.timing_get(KC_STEP)
                ms_s += m1; k_s += k1
            logic.timing_reset()
            b.rollout(3, 0, 512, 0)
            res = b.rollout(3, 0, 512, 0)
            ms_r, k_r = logic.timing_get(KC_ROLLOUT)
            plies_total = sum(r.plies for r in res)
            logic.timing_enable(False)
            sg = {7: 48, 11: 64, 13: 96}[n]
            mask_b = 4 * logic.mask_words
            row = {"config": name, "games": G,
                   "movegen_counts_us": ms_c / k_c * 1e3, "movegen_counts_Mgames_s": G / (ms_c / k_c) / 1e3,
                   "movegen_masks_us": ms_m / k_m * 1e3, "movegen_masks_Mgames_s": G / (ms_m / k_m) / 1e3,
                   "movegen_masks_GBps": G * (sg + 4 + mask_b) / (ms_m / k_m) / 1e6,
                   "step_kth_us": ms_k / k_k * 1e3,
                   "step_us": ms_s / k_s * 1e3, "step_Msteps_s": G / (ms_s / k_s) / 1e3,
                   "step_GBps": G * (2 * sg + 4 + 4 + 40) / (ms_s / k_s) / 1e6,
                   "rollout_ms": ms_r / k_r, "rollout_Gplies_s": plies_total / (ms_r / k_r) / 1e6}
            out.append(row)
            print(json.dumps(row), flush=True)
            b.close()
    return out


if __name__ == "__main__":
    main()
