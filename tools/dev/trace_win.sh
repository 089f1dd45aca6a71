#!/bin/bash
# GPU box: a 2 ms window of the steady state of a bench leg as a per-queue timeline.  usage: trace_win.sh <tag> <env assignments...> -- <bench args>
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for a in "${envs[@]}"; do export "$a"; done
timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python3 $R/bench.py --cpu-frames 0 --batched-streams 0 --gop-parallel 0 --other-configs 0 --regions 1 "$@" > $OUT/bench.json 2> $OUT/bench.err || exit 1
tr=$(find $OUT/tr -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_summary.py $tr > $OUT/summary.txt 2>&1
python3 $R/tools/dev/trace_window.py $tr -12 ${WIN_MS:-2.0} > $OUT/window.txt 2>&1
rm -rf $OUT/tr
python3 -c "import json; d=json.loads(open('$OUT/bench.json').readline()); print('$TAG %.1f M MB/s %.3f ms/step' % (d['value']/1e6, d['ms_per_step']))"
