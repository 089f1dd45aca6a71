#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the default bench line (one 1080p stream, frames pipelined) under the
# knobs of the single-stream schedule, several runs each (the x + y measure once showed a slow mode in 2 of 8 runs).
cd $GRAFT_REPO_ROOT
run() { # label, runs, env...
  local label=$1 n=$2; shift 2
  ( for kv in "$@"; do export "$kv"; done
    printf "%-44s" "$label"
    for i in $(seq $n); do
      timeout -k 10 100 python3 bench.py --cpu-frames 0 --batched-streams 0 --gop-parallel 0 --steps ${STEPS:-300} 2>gpurun_out/sweep_err.txt |
        python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(' %.1f' % (d['value']/1e6), end='')"
    done; echo )
}
run "x+y, 5 stripes, 2 db streams (tree default)" 4
run "8 stripes 3 held, 3 db streams" 6 CEDAR_INTER_BANDS=8 CEDAR_DEFER=3 CEDAR_DEBLOCK_STREAMS=3
run "8 stripes 3 held, 2 db streams" 4 CEDAR_INTER_BANDS=8 CEDAR_DEFER=3 CEDAR_DEBLOCK_STREAMS=2
run "8 stripes 2 held, 3 db streams" 4 CEDAR_INTER_BANDS=8 CEDAR_DEFER=2 CEDAR_DEBLOCK_STREAMS=3
run "6 stripes 2 held, 3 db streams" 4 CEDAR_INTER_BANDS=6 CEDAR_DEFER=2 CEDAR_DEBLOCK_STREAMS=3
run "8 stripes 0 held, 3 db streams" 4 CEDAR_INTER_BANDS=8 CEDAR_DEBLOCK_STREAMS=3
STEPS=20 run "K=20: tree default" 4
STEPS=20 run "K=20: 8 stripes 3 held, 3 db streams" 4 CEDAR_INTER_BANDS=8 CEDAR_DEFER=3 CEDAR_DEBLOCK_STREAMS=3
