cd $GRAFT_REPO_ROOT
B="--cpu-frames 0 --batched-streams 0 --gop-parallel 0"
one() { python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$1: %.1f M  %.3f ms' % (d['value']/1e6, d['ms_per_step']))"; }
run() { env "$@" timeout -k 10 100 python3 bench.py $B --steps 300 2>/dev/null | one "$*"; }
run CEDAR_NOP=1
for nb in 4 6 8 12 16; do run CEDAR_DEBLOCK_STREAMS=3 CEDAR_INTER_BANDS=$nb; done
run CEDAR_DEBLOCK_STREAMS=3
