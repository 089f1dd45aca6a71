cd $GRAFT_REPO_ROOT
B="--cpu-frames 0 --batched-streams 0 --gop-parallel 0"
timeout -k 10 100 python3 bench.py $B --steps 100 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('1080p: %.1f M' % (d['value']/1e6), [(k['kernel'], k['work_ms']) for k in d['roofline']['kernels']])"
timeout -k 10 100 python3 bench.py $B --keyint 1 --steps 60 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('keyint 1: %.1f M MB/s  %.3f ms/step' % (d['value']/1e6, d['ms_per_step']))"
timeout -k 10 100 python3 bench.py $B --width 3840 --height 2160 --steps 100 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('4K: %.1f M' % (d['value']/1e6), [(k['kernel'], k['work_ms']) for k in d['roofline']['kernels']])"
for i in 1 2 3; do timeout -k 10 100 python3 bench.py $B --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('K20: %.1f M MB/s' % (d['value']/1e6))"; done
