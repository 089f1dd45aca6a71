#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): 32 x 1080p lock-step streams under the schedule knobs of the batch
# (cohorts, streams per cohort, planes per deblock workgroup), one bench line each. Development only.
cd $GRAFT_REPO_ROOT
run() { # label, env...
  local label=$1; shift
  ( for kv in "$@"; do export "$kv"; done
    timeout -k 10 120 python3 bench.py --cpu-frames 0 --batched-streams 0 --gop-parallel 0 --streams ${STREAMS:-32} --steps 50 --warmup 25 2>gpurun_out/sweep_err.txt |
      python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-40s %.1f M MB/s  %.3f ms/step' % ('$label', d['value']/1e6, d['ms_per_step']))" )
}
run "default"
run "memset + flag launches" CEDAR_BATCH_SELF_RESET=0
run "planes split" CEDAR_DEBLOCK_PLANES=1
run "3 cohorts" CEDAR_COHORTS=3
run "default again"
STREAMS=8 run "8 streams"
STREAMS=16 run "16 streams"
STREAMS=16 run "16 streams, 2 cohorts" CEDAR_COHORTS=2
STREAMS=64 run "64 streams"
