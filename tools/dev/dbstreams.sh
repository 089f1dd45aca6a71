cd $GRAFT_REPO_ROOT
B="--cpu-frames 0 --batched-streams 0 --gop-parallel 0"
one() { python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$1: %.1f M  %.3f ms' % (d['value']/1e6, d['ms_per_step']), d['roofline']['schedule'])"; }
for n in 2 3; do
  CEDAR_DEBLOCK_STREAMS=$n timeout -k 10 100 python3 bench.py $B --steps 300 2>/dev/null | one "1 stream, $n deblock streams"
  CEDAR_DEBLOCK_STREAMS=$n CEDAR_WF_SPLIT=4 timeout -k 10 100 python3 bench.py $B --steps 300 2>/dev/null | one "1 stream, $n deblock streams, split 4"
  CEDAR_DEBLOCK_STREAMS=$n timeout -k 10 100 python3 bench.py $B --steps 100 --streams 2 2>/dev/null | one "2 streams, $n deblock streams"
done
