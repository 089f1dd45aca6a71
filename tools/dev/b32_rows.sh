#!/bin/bash
# GPU box: the big batches on k_deblock_rows now that the entropy chain is shorter. Development.
cd $GRAFT_REPO_ROOT
run() { local label=$1 S=$2; shift 2
  ( for kv in "$@"; do export "$kv"; done
    BENCH_NO_PROFILE=1 timeout -k 10 120 python3 bench.py --cpu-frames 0 --batched-streams 0 --gop-parallel 0 --other-configs 0 --regions 1 --streams $S --steps 50 --warmup 25 2>gpurun_out/sweep_err.txt |
      python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-56s S=%-3s %.1f M MB/s  %.3f ms/step' % ('$label', '$S', d['value']/1e6, d['ms_per_step']))" )
}
for S in 24 32 48 64; do
  run "default" $S
  run "k_deblock_rows" $S CEDAR_DEBLOCK_ROWS=2
  run "k_deblock_rows, 1 cohort" $S CEDAR_DEBLOCK_ROWS=2 CEDAR_COHORTS=1
  run "k_deblock_rows, 2 cohorts, 6 in flight" $S CEDAR_DEBLOCK_ROWS=2 CEDAR_COHORTS=2 CEDAR_PIPE_DEPTH=6
done
