# the default bench under rocprofv3 --kernel-trace until a run fails; keeps the failing run's kernel trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
n=0; runs=${1:-30}
for i in $(seq 1 $runs); do
  rm -rf /tmp/ht
  CEDAR_DEBUG_HANG=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/ht -- python3 $R/bench.py --cpu-frames 0 --batched-streams 0 --gop-parallel 0 --steps 100 > $R/gpurun_out/ht.json 2> $R/gpurun_out/ht.err
  rc=$?
  echo "run $i rc $rc"
  if [ $rc -ne 0 ]; then
    cp $(find /tmp/ht -name "*kernel_trace.csv" | head -1) $R/gpurun_out/hang_kernel_trace.csv
    grep -A1 HANG $R/gpurun_out/ht.err | cut -c1-300
    break
  fi
done
