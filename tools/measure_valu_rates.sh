#!/bin/sh
# builds tools/valu_rates.hip for gfx950 and prints the issue cost of the playout kernel's instructions with 1 / 2 / 4 waves per SIMD
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o /tmp/valu_rates valu_rates.hip
for w in 1 2 4; do timeout -k 5 60 /tmp/valu_rates $w; done
