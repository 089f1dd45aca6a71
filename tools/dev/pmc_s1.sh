#!/bin/bash
# GPU box: the two PMC passes (FETCH_SIZE, WRITE_SIZE) of one 1080p stream, frames back to back -> gpurun_out/<tag>/pmc_s1_*.csv
set -o pipefail
TAG=${1:-pmc1}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
export CEDAR_ROW_PIPE=0
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 280 rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_s1_$ctr -- python3 $R/bench.py --cpu-frames 0 --batched-streams 0 --gop-parallel 0 --other-configs 0 --regions 1 --steps 25 --warmup 25 > $OUT/pmc_s1_$ctr.json 2> $OUT/pmc_s1_$ctr.err || exit 1
  cp $(find $OUT/pmc_s1_$ctr -name "*counter_collection.csv" | head -1) $OUT/pmc_s1_$ctr.csv; rm -rf $OUT/pmc_s1_$ctr
done
echo collected $TAG
