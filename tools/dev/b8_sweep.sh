#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): 8 (or $STREAMS) x 1080p lock-step streams, frames back to back on k_deblock_rows, under the
# cohort knobs -- cohorts as independent chains: each cohort's deblocking (and entropy) on a stream of its own. Development only.
cd $GRAFT_REPO_ROOT
run() { # label, env...
  local label=$1; shift
  ( for kv in "$@"; do export "$kv"; done
    timeout -k 10 120 python3 bench.py --cpu-frames 0 --batched-streams 0 --gop-parallel 0 --other-configs 0 --regions 1 --streams ${STREAMS:-8} --steps 50 --warmup 25 2>gpurun_out/sweep_err.txt |
      python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-64s %.1f M MB/s  %.3f ms/step' % ('$label', d['value']/1e6, d['ms_per_step']))" )
}
run "default"
run "2 cohorts" CEDAR_COHORTS=2
run "2 cohorts, deblock streams" CEDAR_COHORTS=2 CEDAR_COHORT_STREAMS=1
run "2 cohorts, deblock + entropy streams" CEDAR_COHORTS=2 CEDAR_COHORT_STREAMS=3
run "2 cohorts, deblock + entropy streams, self-reset" CEDAR_COHORTS=2 CEDAR_COHORT_STREAMS=3 CEDAR_BATCH_SELF_RESET=1
run "4 cohorts, deblock + entropy streams" CEDAR_COHORTS=4 CEDAR_COHORT_STREAMS=3
run "4 cohorts, deblock + entropy streams, self-reset" CEDAR_COHORTS=4 CEDAR_COHORT_STREAMS=3 CEDAR_BATCH_SELF_RESET=1
run "default, round-3 back end and k_input" CEDAR_ENTROPY_FUSED=0 CEDAR_SRC_DIRECT=0
run "default again"
