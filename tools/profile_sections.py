#!/usr/bin/env python3
"""In-kernel section timers of the playout loop ("Where a ply goes").

Needs a PROFILING build of the library (never the product build):
  cd alphazeroforhnefatafl_amd/csrc && hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math \
      -mllvm --amdgpu-sched-strategy=max-ilp -DTAFL_PROF -shared -o /tmp/libtaflhip_prof.so tafl_capi.hip tafl_replay.cpp
  TAFLHIP_PROF_LIB=/tmp/libtaflhip_prof.so python tools/profile_sections.py [c11|b7|c13]
Sections are bracketed by s_memtime reads behind scheduling barriers (TAFL_PROF_* in tafl_bits.hpp), one lane per wave adding into
its workgroup's row; two empty sections give the cost of a mark.  The optimiser still moves straight-line code across marks, so
only coarse sections (pick / apply / T upkeep / gen / outcome) are meaningful."""
import ctypes as C, os, sys
here = os.path.dirname(os.path.abspath(__file__))
os.environ['TAFLHIP_LIB'] = os.environ.get('TAFLHIP_PROF_LIB', '/tmp/libtaflhip_prof.so')
sys.path.insert(0, os.path.dirname(here))
from alphazeroforhnefatafl_amd import BatchedGameLogic, rules, boards
from alphazeroforhnefatafl_amd._lib import lib
names = ['rng+pick', 'move+fields+custodial', 'king-adjacent', 'shieldwall filter', 'rest of apply_pre', 'T upkeep', 'gen', 'outcome+finish']
board = sys.argv[1] if len(sys.argv) > 1 else 'c11'
if board == 'c11': lg = BatchedGameLogic(rules.COPENHAGEN, 11); fen = boards.COPENHAGEN
elif board == 'b7': lg = BatchedGameLogic(rules.BRANDUBH, 7); fen = boards.BRANDUBH
else: lg = BatchedGameLogic(rules.COPENHAGEN, 13); fen = boards.COPENHAGEN13
b = lg.new_batch(65536, fen)
L = lib()
L.tafl_prof_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]; L.tafl_prof_read.restype = C.c_int
out = (C.c_ulonglong * 32)()
b.rollout(1, 0, 512); lg.sync(); L.tafl_prof_read(out, 1)
for sim in range(1, 4): b.rollout(1, sim, 512)
lg.sync(); L.tafl_prof_read(out, 1)
n = out[31]
tot = 0
vals = [(out[k] if out[k] < 2**63 else out[k] - 2**64) for k in range(8)]
ov = [(out[k] if out[k] < 2**63 else out[k] - 2**64) / n for k in (10, 11)]
print('empty sections (mark overhead):', ov)
tot = sum(vals)
print(board, 'wave-plies', n, 'cycles/ply', round(tot / n, 1))
for k in range(8): print('  %-24s %7.1f cycles/ply  net %7.1f' % (names[k], vals[k] / n, vals[k] / n - ov[0]))
for k, nm in ((8, 'enclosure flood (in outcome)'), (9, 'exit fort (in outcome)')):
    v = out[k] if out[k] < 2**63 else out[k] - 2**64
    print('  %-30s %7.1f cycles/ply' % (nm, v / n))
