#!/bin/bash
# Runs ON THE GPU BOX (through gpurun). Round-4 evidence:
#   w1         one 1080p stream, frames back to back (CEDAR_ROW_PIPE=0): kernel durations = work
#   p1         the same stream as bench.py runs it (frames pipelined, IDR look-ahead)
#   s8 / s32   8 and 32 lock-step streams
#   g8         ONE stream, 8 closed GOPs side by side (gop_parallel)
# Each leg: rocprofv3 --kernel-trace --stats of bench.py, the stats CSV and a text summary of
# the trace (tools/trace_summary.py). Then the PMC passes (FETCH_SIZE and WRITE_SIZE separately, as
# the guide prescribes, frames back to back: counter collection serialises dispatches) for 1 and
# 32 streams. Outputs under gpurun_out/<tag>/; tools/publish_r04.py copies the summaries to profiles/.
set -o pipefail
TAG=${1:-r04a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
cd /tmp
leg() { # name, env assignment (may be empty), bench args...
  local name=$1 envs=$2; shift 2
  ( [ -n "$envs" ] && export $envs
    timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- \
      python3 $R/bench.py --cpu-frames 0 --batched-streams 0 --gop-parallel 0 --other-configs 0 "$@" > $OUT/$name.json 2> $OUT/$name.err ) || return 1
  local tr=$(find $OUT/$name -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/trace_summary.py $tr > $OUT/${name}_trace_summary.txt 2>&1
  cp $(find $OUT/$name -name "*kernel_stats.csv" | head -1) $OUT/${name}_kernel_stats.csv
  rm -rf $OUT/$name
  echo "leg $name done"
}
pmc() { # name, counter, bench args
  local name=$1 ctr=$2; shift 2
  ( export CEDAR_ROW_PIPE=0
    timeout -k 10 280 rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_${name}_$ctr -- \
      python3 $R/bench.py --cpu-frames 0 --batched-streams 0 --gop-parallel 0 --other-configs 0 --steps 25 --warmup 25 "$@" > $OUT/pmc_${name}_$ctr.json 2> $OUT/pmc_${name}_$ctr.err ) || return 1
  cp $(find $OUT/pmc_${name}_$ctr -name "*counter_collection.csv" | head -1) $OUT/pmc_${name}_$ctr.csv
  rm -rf $OUT/pmc_${name}_$ctr
  echo "pmc $name $ctr done"
}
if [ -n "$LEGS" ]; then # (only the kernel-trace legs named, e.g. LEGS="w1 p1")
  for l in $LEGS; do
    case $l in
      w1) leg w1 CEDAR_ROW_PIPE=0 --steps 50 --warmup 25;;
      p1) leg p1 "" --steps 100 --warmup 25;;
      s8) leg s8 "" --streams 8 --steps 50 --warmup 25;;
      s32) leg s32 "" --streams 32 --steps 50 --warmup 25;;
    esac || exit 1
  done
  echo collected $TAG; exit 0
fi
leg w1 CEDAR_ROW_PIPE=0 --steps 50 --warmup 25 &&
leg p1 "" --steps 100 --warmup 25 &&
leg s8 "" --streams 8 --steps 50 --warmup 25 &&
leg s32 "" --streams 32 --steps 50 --warmup 25 &&
pmc s1 FETCH_SIZE && pmc s1 WRITE_SIZE &&
pmc s32 FETCH_SIZE --streams 32 && pmc s32 WRITE_SIZE --streams 32 &&
echo collected $TAG
