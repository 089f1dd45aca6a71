# runs on the GPU box: the table of BASELINE.md section 5 that is not in profiles/ (development aid)
cd $GRAFT_REPO_ROOT
B="--cpu-frames 0 --batched-streams 0 --gop-parallel 0"
one() { python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1: %.1f M MB/s  %.3f ms/step' % (d['value']/1e6, d['ms_per_step']))"; }
timeout -k 10 200 python3 bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err && head -c 400 gpurun_out/r02_bench_default.json && echo
timeout -k 10 100 python3 bench.py $B --steps 20 --warmup 5 | one "driver args K20 W5"
for s in 8 16 32 64; do timeout -k 10 100 python3 bench.py $B --streams $s --steps 50 --warmup 25 | one "streams $s"; done
timeout -k 10 100 python3 bench.py $B --width 854 --height 480 --qp 30 | one "480p"
timeout -k 10 100 python3 bench.py $B --width 3840 --height 2160 --steps 100 | one "4K"
timeout -k 10 300 python3 tools/gop_bench.py 4 8 16 32
