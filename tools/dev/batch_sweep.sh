#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): 32 x 1080p lock-step streams under the schedule knobs of the batch
# (cohorts, streams per cohort, self-resetting kernels, planes per deblock workgroup), one bench line each. Development only.
cd $GRAFT_REPO_ROOT
run() { # label, env...
  local label=$1; shift
  ( for kv in "$@"; do export "$kv"; done
    timeout -k 10 120 python3 bench.py --cpu-frames 0 --batched-streams 0 --gop-parallel 0 --other-configs 0 --streams ${STREAMS:-32} --steps 50 --warmup 25 2>gpurun_out/sweep_err.txt |
      python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-52s %.1f M MB/s  %.3f ms/step' % ('$label', d['value']/1e6, d['ms_per_step']))" )
}
run "default"
run "self-zero + cohort deblock streams" CEDAR_BATCH_SELF_RESET=1 CEDAR_COHORT_STREAMS=1
run "self-zero + cohort deblock and entropy streams" CEDAR_BATCH_SELF_RESET=1 CEDAR_COHORT_STREAMS=3
run "3 cohorts, self-zero + deblock streams" CEDAR_BATCH_SELF_RESET=1 CEDAR_COHORT_STREAMS=1 CEDAR_COHORTS=3
run "3 cohorts, self-zero + deblock and entropy streams" CEDAR_BATCH_SELF_RESET=1 CEDAR_COHORT_STREAMS=3 CEDAR_COHORTS=3
run "default again"
