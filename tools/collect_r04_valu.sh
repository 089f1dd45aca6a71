#!/bin/bash
# Runs ON THE GPU BOX (through gpurun). Round-4 evidence behind "VALU-bound": SQ counters per kernel.
#   w1   one 1080p stream, frames back to back (CEDAR_ROW_PIPE=0)
#   s32  32 lock-step streams (two cohorts)
# Two passes per leg (the SQ block has 8 slots, GRBM 2; the passes are kept small so that no counter is multiplexed):
#   A  SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS
#   B  SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
# The python program itself follows `--` (no env / bash -c hop: the profiler's library initialises the GPU first).
# Outputs under gpurun_out/<tag>/; tools/pmc_valu_summary.py turns them into profiles/r04_{w1,s32}_pmc_valu.csv.
set -o pipefail
TAG=${1:-r04valu}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
export CEDAR_ROW_PIPE=0
pmc() { # name, pass, counters, bench args
  local name=$1 pass=$2 ctrs=$3; shift 3
  timeout -k 10 280 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc_${name}_$pass -- \
      python3 $R/bench.py --cpu-frames 0 --batched-streams 0 --gop-parallel 0 --other-configs 0 --steps 25 --warmup 25 "$@" > $OUT/pmc_${name}_$pass.json 2> $OUT/pmc_${name}_$pass.err || return 1
  cp $(find $OUT/pmc_${name}_$pass -name "*counter_collection.csv" | head -1) $OUT/pmc_${name}_$pass.csv
  rm -rf $OUT/pmc_${name}_$pass
  echo "pmc $name $pass done"
}
pmc w1 A "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS" &&
pmc w1 B "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" &&
pmc s32 A "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS" --streams 32 &&
pmc s32 B "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" --streams 32 &&
echo collected $TAG
