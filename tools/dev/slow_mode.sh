#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the default bench N times (one stream, K 300 and the driver's K 20) with what the
# session says of its schedule, and for a slow run the CONFIG line (CEDAR_TRACE_SUBMIT): how often does a run land in a slow
# schedule, and what had CONFIG found then?
cd $GRAFT_REPO_ROOT
N=${1:-12}
for i in $(seq 1 $N); do
  for K in "" "--steps 20 --warmup 5"; do
    CEDAR_TRACE_SUBMIT=1 timeout -k 10 120 python3 bench.py $K --cpu-frames 0 --batched-streams 0 --gop-parallel 0 --other-configs 0 2>gpurun_out/slow_err.txt | tee gpurun_out/slow_last.json |
      python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); v=d['value']/1e6; s=(d.get('roofline') or {}).get('schedule')
print('run $i [$K]: %.1f M  regions %s  %s' % (v, ' '.join('%.1f' % (x/1e6) for x in d.get('region_runs', [])), s))
sys.exit(3 if v < 45 else 0)"
    if [ $? -eq 3 ]; then grep "^config:" gpurun_out/slow_err.txt; cp gpurun_out/slow_err.txt gpurun_out/slow_err_$i.txt; cp gpurun_out/slow_last.json gpurun_out/slow_json_$i.json; else cp gpurun_out/slow_last.json gpurun_out/slow_json_normal.json; fi
  done
done
