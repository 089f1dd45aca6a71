# usage: trace_one.sh <tag> <env assignments...> -- <bench args>   (runs on the GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for a in "${envs[@]}"; do export "$a"; done
timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python3 $R/bench.py --cpu-frames 0 --batched-streams 0 --gop-parallel 0 "$@" > $OUT/bench.json 2> $OUT/bench.err || exit 1
tr=$(find $OUT/tr -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_summary.py $tr > $OUT/summary.txt 2>&1
python3 $R/tools/dev/trace_timeline.py $tr ${TL_N:-120} ${TL_MIN:-5} > $OUT/timeline.txt 2>&1
rm -rf $OUT/tr
