# sweeps behind DESIGN 5.2 (CEDAR_DEFER only exists with tools/dev/xy_defer.patch applied)
cd $GRAFT_REPO_ROOT
B="--cpu-frames 0 --batched-streams 0 --gop-parallel 0"
one() { python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$1: %.1f M  %.3f ms' % (d['value']/1e6, d['ms_per_step']))"; }
run() { env "$@" timeout -k 10 100 python3 bench.py $B --steps 300 2>/dev/null | one "$*"; }
run CEDAR_NOP=1
run CEDAR_DEFER=2 CEDAR_DEBLOCK_STREAMS=3 CEDAR_INTER_BANDS=6
run CEDAR_DEFER=3 CEDAR_DEBLOCK_STREAMS=3 CEDAR_INTER_BANDS=8
run CEDAR_DEFER=4 CEDAR_DEBLOCK_STREAMS=3 CEDAR_INTER_BANDS=8
run CEDAR_DEFER=4 CEDAR_DEBLOCK_STREAMS=3 CEDAR_INTER_BANDS=10
run CEDAR_DEFER=3 CEDAR_DEBLOCK_STREAMS=2 CEDAR_INTER_BANDS=8
