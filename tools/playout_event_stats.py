#!/usr/bin/env python3
"""How often a game takes each rare path of the playout loop: the device code compiled for the HOST with
event counters (-DTAFL_STAT), 3 000 random playouts from the start position.  CPU only.

  g++ -O2 -std=c++17 -fPIC -ffp-contract=off -DTAFL_STAT -shared -o /tmp/libhostsim_stat.so tests/hostsim/hostsim.cpp -lm
  python tools/playout_event_stats.py [c11|b7]
P(wave of 64) = 1 - (1 - p)^64: the share of wave-plies in which at least one of 64 games takes the path."""
import ctypes as C, sys
ROOT = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/oracle')
from alphazeroforhnefatafl_amd import abi
from alphazeroforhnefatafl_amd.abi import TaflRules, TaflState, TaflRolloutResult
import oracle as orc
import os
L = C.CDLL(os.environ.get('TAFL_HOSTSIM_STAT_LIB', '/tmp/libhostsim_stat.so'))
acc = (C.c_ulonglong * 32).in_dll(L, 'tafl_stat_acc')
board = sys.argv[1] if len(sys.argv) > 1 else 'c11'
if board == 'c11': rules, n, fen, wb = abi.rules.COPENHAGEN, 11, abi.boards.COPENHAGEN, 128
else: rules, n, fen, wb = abi.rules.BRANDUBH, 7, abi.boards.BRANDUBH, 64
G = 3000
st = orc.GameState(fen, rules.starting_side, wb).to_abi()
states = (TaflState * G)(*([st] * G))
out = (TaflRolloutResult * G)()
r = rules.to_c()
L.hs_rollout.argtypes = [C.POINTER(TaflRules), C.c_uint8, C.c_uint32, C.POINTER(TaflState), C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(TaflRolloutResult)]
assert L.hs_rollout(C.byref(r), n, wb, states, G, 2, 0, 512, 0, out) == 0
N = acc[31]
names = {0: 'exit_fort entered (defender moved)', 1: 'king on edge', 2: 'ring1: no attacker next to king', 3: 'ring1 empty nb, no corner', 4: 'ring2 passed -> flood', 5: 'flood ok -> secure',
         13: 'fort candidate (edge-line flank test)', 8: 'shieldwall filter passed', 9: 'king adjacent', 10: 'enclosure flood entered', 11: 'ply with captures', 12: 'non-custodial captures'}
print(board, 'plies', N)
for k, nm in names.items():
    p = acc[k] / N
    print('  %-38s %9d  %.4f %% per ply   P(wave of 64) = %.3f' % (nm, acc[k], 100 * p, 1 - (1 - p) ** 64))
