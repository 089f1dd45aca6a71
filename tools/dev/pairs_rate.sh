cd $GRAFT_REPO_ROOT
B="--cpu-frames 0 --batched-streams 0 --gop-parallel 0"
one() { python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$1: %.1f M' % (d['value']/1e6), [(k['kernel'], k['work_ms']) for k in d['roofline']['kernels'] if 'deblock' in k['kernel']], d['roofline'].get('wf_split'))"; }
for pr in 8 4 2; do
  CEDAR_DB_PAIRS=$pr timeout -k 10 100 python3 bench.py $B --steps 200 2>/dev/null | one "1 stream pairs $pr"
  CEDAR_DB_PAIRS=$pr timeout -k 10 100 python3 bench.py $B --steps 100 --streams 4 2>/dev/null | one "4 streams pairs $pr"
  CEDAR_DB_PAIRS=$pr timeout -k 10 100 python3 bench.py $B --steps 100 --streams 8 2>/dev/null | one "8 streams pairs $pr"
done
CEDAR_DB_PAIRS=4 CEDAR_WF_SPLIT=8 timeout -k 10 100 python3 bench.py $B --steps 200 2>/dev/null | one "1 stream pairs 4 split 8"
