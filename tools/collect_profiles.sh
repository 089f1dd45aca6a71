#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): default bench, rocprofv3 kernel trace + stats,
# and the two PMC passes (FETCH_SIZE, WRITE_SIZE separately, as the guide prescribes).
# Outputs land in gpurun_out/<tag>/; tools/pmc_summary.py + a copy step turn them
# into profiles/<tag>_* afterwards.
set -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout -k 10 500 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- \
    python3 $R/bench.py --steps 50 --warmup 25 --batched-streams 0 --cpu-frames 0 > $OUT/trace_bench.json 2> $OUT/trace.err || exit 2
# counter collection serialises kernel dispatches: kernels that poll each other's progress
# (frame pipelining) would wait for their time-out, so the PMC passes run frames back to back
export CEDAR_ROW_PIPE=0
for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- \
        python3 $R/bench.py --steps 25 --warmup 25 --batched-streams 0 --cpu-frames 0 > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err || exit 3
done
cd $R
find $OUT -name "*kernel_trace.csv" -delete   # large; the stats summary is what is kept
echo collected $TAG
