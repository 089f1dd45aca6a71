# distribution of the default bench line over several runs, with the stream roles each session was dealt
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8; do CEDAR_TRACE_SUBMIT=1 timeout -k 10 100 python3 bench.py --cpu-frames 0 --batched-streams 0 --gop-parallel 0 --steps 300 2>gpurun_out/err_$i.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('run: %.1f M MB/s  %.3f ms/step' % (d['value']/1e6, d['ms_per_step']))"; grep -m1 "roles" gpurun_out/err_$i.txt | sed 's/.*roles/roles/'; done
