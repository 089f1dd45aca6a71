#!/bin/bash
# Runs ON THE GPU BOX: the default bench line N times with the stream roles each measured session was dealt
# (the slow mode of the single-stream schedule: which assignment, which measurement?). Development only.
cd $GRAFT_REPO_ROOT
for i in $(seq ${1:-12}); do
  CEDAR_TRACE_SUBMIT=1 timeout -k 10 100 python3 bench.py --cpu-frames 0 --batched-streams 0 --gop-parallel 0 --steps 300 2>gpurun_out/roles_err_$i.txt |
    python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('run: %.1f M MB/s  %.3f ms/step' % (d['value']/1e6, d['ms_per_step']))"
  grep "roles" gpurun_out/roles_err_$i.txt | sed 's/.*extra us per launch (blocked -> chain)://' | tail -2
  rm -f gpurun_out/roles_err_$i.txt
done
