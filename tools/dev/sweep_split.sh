cd $GRAFT_REPO_ROOT
run() { # env..., streams from $ST
  r=$(env "$@" timeout -k 10 120 python3 bench.py --steps 50 --warmup 25 --cpu-frames 0 --batched-streams 0 --gop-parallel 0 --streams $ST 2>&1 | python3 -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('%.1f M  %.3f ms' % (d['value']/1e6, d['ms_per_step'])); break
else: print('failed')")
  echo "streams $ST $*: $r"
}
for ST in 12 16 24; do
  run CEDAR_NOP=1
  run CEDAR_ROW_PIPE=1 CEDAR_PIPE_DEPTH=8 CEDAR_DB_PAIRS=2
  run CEDAR_ROW_PIPE=1 CEDAR_PIPE_DEPTH=8 CEDAR_DB_PAIRS=8 CEDAR_WF_SPLIT=2
  run CEDAR_ROW_PIPE=1 CEDAR_PIPE_DEPTH=6 CEDAR_DB_PAIRS=2 CEDAR_DEBLOCK_PLANES=0
done
