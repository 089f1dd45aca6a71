cd $GRAFT_REPO_ROOT
run() { # env..., streams from $ST
  r=$(env "$@" timeout -k 10 120 python3 bench.py --steps 40 --warmup 10 --cpu-frames 0 --batched-streams 0 --gop-parallel 0 --streams $ST 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.1f M  %.3f ms' % (d['value']/1e6, d['ms_per_step']))")
  echo "streams $ST $*: $r"
}
for ST in 32 16; do
  run CEDAR_DB_PAIRS=2 CEDAR_DB_RELAX=0
  run CEDAR_DB_PAIRS=2 CEDAR_DB_RELAX=1
  run CEDAR_DB_PAIRS=2 CEDAR_COHORTS=2 CEDAR_DEBLOCK_PLANES=0 CEDAR_DB_RELAX=0
  run CEDAR_DB_PAIRS=2 CEDAR_COHORTS=2 CEDAR_DEBLOCK_PLANES=0 CEDAR_DB_RELAX=1
  run CEDAR_DB_PAIRS=8 CEDAR_DB_RELAX=1
done
