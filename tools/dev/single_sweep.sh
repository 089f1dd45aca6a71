#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the default bench line (one stream, frames pipelined), several runs each
# (the x + y measure once showed a slow mode in 2 of 8 runs), at the sizes of BASELINE configs[1..3].
cd $GRAFT_REPO_ROOT
run() { # label, runs, bench args..., then env after --
  local label=$1 n=$2; shift 2
  local args=()
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done
  [ "$1" = "--" ] && shift
  ( for kv in "$@"; do export "$kv"; done
    printf "%-44s" "$label"
    for i in $(seq $n); do
      timeout -k 10 100 python3 bench.py --cpu-frames 0 --batched-streams 0 --gop-parallel 0 "${args[@]}" 2>gpurun_out/sweep_err.txt |
        python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(' %.1f' % (d['value']/1e6), end='')"
    done; echo )
}
run "1080p default (K 300)" ${RUNS:-10}
run "1080p K=20 W=5" 6 --steps 20 --warmup 5
run "1080p CEDAR_DEFER=0" 3 -- CEDAR_DEFER=0
run "854x480 default" 3 --width 854 --height 480 --qp 30
run "854x480 6 stripes 2 held 3 db" 3 --width 854 --height 480 --qp 30 -- CEDAR_INTER_BANDS=6 CEDAR_DEFER=2
run "854x480 8 stripes 3 held 3 db" 3 --width 854 --height 480 --qp 30 -- CEDAR_INTER_BANDS=8 CEDAR_DEFER=3
run "3840x2160 default" 3 --width 3840 --height 2160
run "3840x2160 CEDAR_DEFER=0" 3 --width 3840 --height 2160 -- CEDAR_DEFER=0
