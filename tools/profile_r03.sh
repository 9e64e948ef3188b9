#!/bin/bash
# One profiling session on the GPU box (gpurun): the default bench line, the kernel trace of the same command and three separate PMC
# passes (FETCH_SIZE / WRITE_SIZE / SQ counters; never combined with API traces).  Outputs under gpurun_out/$1 (default r03); the
# summaries that are kept go to profiles/ (tools/summarize_profiles.py).
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r03}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/kt.log 2>&1
echo "kt done"
# the same kernel trace for the headline workload alone (no variants): its per-kernel averages are the ones to compare with the HIP-event
# averages in that run's own bench line; the raw trace is kept (overlap of the partitions' launches)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_headline -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-variants > $O/kt_headline.log 2>&1
echo "kt headline done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-variants > $O/fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-variants > $O/write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-variants > $O/sq.log 2>&1
echo "sq done"
