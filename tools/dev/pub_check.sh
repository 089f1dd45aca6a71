#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the whole default bench and two 1080p streams with a band's progress word every 8 and
# every 16 steps of k_deblock_rows (CEDAR_DB_PUB_STEP). Development only.
cd $GRAFT_REPO_ROOT
for P in ${VALS:-8 16 8 16}; do
  echo "== ${VAR:-CEDAR_DB_PUB_STEP}=$P"
  ( export ${VAR:-CEDAR_DB_PUB_STEP}=$P
    timeout -k 10 300 python3 bench.py --cpu-frames 0 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('one stream %.1f M (regions %s); batched %s; GOP lanes %s; configs %s' % (d['value']/1e6, ' '.join('%.1f' % (v/1e6) for v in d.get('region_runs', [])),
  ' / '.join('%d: %.1f' % (b['streams_per_gpu'], b['value']/1e6) for b in d.get('batched', [])),
  ' / '.join('%d: %.1f' % (g['gops_side_by_side'], g['value']/1e6) for g in d.get('gop_parallel', [])),
  ' / '.join('%.1f' % (c['value']/1e6) for c in d.get('configs', []))))"
    timeout -k 10 300 python3 bench.py --streams 2 --cpu-frames 0 --batched-streams 0 --gop-parallel 0 --other-configs 0 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('2 streams: %.1f M (regions %s)' % (d['value']/1e6, ' '.join('%.1f' % (v/1e6) for v in d.get('region_runs', []))))" )
done
