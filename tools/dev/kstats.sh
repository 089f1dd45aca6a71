#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of one bench leg.  usage: kstats.sh <name> [env assignment or ""] bench args...
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/kstats
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
name=$1; envs=$2; shift 2
( [ -n "$envs" ] && export $envs
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- \
    python3 $R/bench.py --cpu-frames 0 --batched-streams 0 --gop-parallel 0 --other-configs 0 --regions 1 "$@" > $OUT/$name.json 2> $OUT/$name.err ) || exit 1
cp $(find $OUT/$name -name "*kernel_stats.csv" | head -1) $OUT/${name}_kernel_stats.csv
python3 $R/tools/trace_summary.py $(find $OUT/$name -name "*kernel_trace.csv" | head -1) > $OUT/${name}_trace_summary.txt 2>&1
rm -rf $OUT/$name
cut -d, -f1-4 $OUT/${name}_kernel_stats.csv | head -24
