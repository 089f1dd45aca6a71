#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the thread-per-macroblock k_cavlc with 64 / 32 / 16 / 8 macroblocks to a wave
# (CEDAR_CAVLC_LANES), at 8, 16 and 32 lock-step 1080p streams. Development only.
cd $GRAFT_REPO_ROOT
run() { # streams, label, env...
  local streams=$1 label=$2; shift 2
  ( for kv in "$@"; do export "$kv"; done
    timeout -k 10 120 python3 bench.py --cpu-frames 0 --batched-streams 0 --gop-parallel 0 --other-configs 0 --regions 1 --streams $streams --steps 50 --warmup 25 2>gpurun_out/sweep_err.txt |
      python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%2d streams, %-40s %.1f M MB/s  %.3f ms/step' % ($streams, '$label', d['value']/1e6, d['ms_per_step']))" )
}
for S in ${STREAMS_LIST:-8 16 32}; do
  run $S "64 macroblocks a wave (default)"
  run $S "32" CEDAR_CAVLC_LANES=32
  run $S "16" CEDAR_CAVLC_LANES=16
  run $S "8" CEDAR_CAVLC_LANES=8
  run $S "64 again"
done
