#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): one 1080p stream, default run (K 300) and the driver's arguments (K 20), under
# schedule knobs given as "label|ENV=.. ENV=.." lines on stdin. Development only.
cd $GRAFT_REPO_ROOT
while IFS='|' read -r label envs; do
  [ -z "$label" ] && continue
  ( for kv in $envs; do export "$kv"; done
    a=$(timeout -k 10 120 python3 bench.py --cpu-frames 0 --batched-streams 0 --gop-parallel 0 --other-configs 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.1f' % (d['value']/1e6))")
    b=$(timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --cpu-frames 0 --batched-streams 0 --gop-parallel 0 --other-configs 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.1f' % (d['value']/1e6))")
    echo "$label: K300 $a M  K20 $b M" )
done
