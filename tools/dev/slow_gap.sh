#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): does the slow run depend on how soon a process starts after the one
# before it has ended? N runs of the one-stream bench back to back, then N with a pause in front of each. Development only.
cd $GRAFT_REPO_ROOT
N=${1:-40}
for gap in 0 3; do
  slow=0
  for i in $(seq 1 $N); do
    [ $gap -gt 0 ] && sleep $gap
    v=$(timeout -k 10 120 python3 bench.py --cpu-frames 0 --batched-streams 0 --gop-parallel 0 --other-configs 0 --regions 1 2>/dev/null |
      python3 -c "import sys,json; print('%.1f' % (json.loads(sys.stdin.readline())['value']/1e6))")
    if python3 -c "import sys; sys.exit(0 if float('$v') < 45 else 1)"; then slow=$((slow + 1)); echo "pause $gap s, run $i: $v M  SLOW"; fi
  done
  echo "pause $gap s: $slow slow of $N"
done
