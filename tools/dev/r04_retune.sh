#!/bin/bash
# GPU box: the plan's thresholds again after the round's back end: stripes of one stream, which deblock kernel for 3 to 6 streams. Development.
cd $GRAFT_REPO_ROOT
run() { local label=$1 S=$2 K=$3; shift 3
  ( for kv in "$@"; do export "$kv"; done
    BENCH_NO_PROFILE=1 timeout -k 10 120 python3 bench.py --cpu-frames 0 --batched-streams 0 --gop-parallel 0 --other-configs 0 --regions 3 --streams $S --steps $K --warmup 25 2>gpurun_out/sweep_err.txt |
      python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-44s S=%-2s K=%-3s %s M MB/s' % ('$label', '$S', '$K', ' '.join('%.1f' % (x/1e6) for x in d['region_runs'])))" )
}
for nb in 2 3 4 5; do run "stripes $nb" 1 300 CEDAR_INTER_BANDS=$nb; run "stripes $nb" 1 20 CEDAR_INTER_BANDS=$nb; done
for S in 3 4 5 6; do
  run "default" $S 100
  run "k_deblock_rows, frames back to back" $S 100 CEDAR_DEBLOCK_ROWS=2 CEDAR_ROW_PIPE=0
  run "wave pairs, pipelined" $S 100 CEDAR_DEBLOCK_ROWS=0
done
