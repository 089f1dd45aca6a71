#!/bin/bash
# GPU box: how k_pack_out's bytes should reach the host. Development.
cd $GRAFT_REPO_ROOT
run() { local label=$1; shift
  ( for kv in "$@"; do export "$kv"; done
    timeout -k 10 60 python3 tools/dev/pk_flaky.py 6 2>&1 | tail -1 | tr '\n' ' '
    for S in 1 8 32; do
      timeout -k 10 120 python3 bench.py --cpu-frames 0 --batched-streams 0 --gop-parallel 0 --other-configs 0 --regions 1 --streams $S --steps 50 --warmup 25 2>gpurun_out/sweep_err.txt |
        python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(' S=$S %.1f M' % (d['value']/1e6), end='')"
    done; echo "   <- $label" )
}
run "plain stores + release fence per workgroup (mode 2)" CEDAR_PK_STORE=2
run "coherent host memory, plain stores (mode 0)" CEDAR_PK_STORE=0 CEDAR_HOST_COHERENT=1
run "coherent host memory, mode 2" CEDAR_PK_STORE=2 CEDAR_HOST_COHERENT=1
run "non-coherent host memory, plain stores, no fence (wrong?)" CEDAR_PK_STORE=0
run "round-3 back end" CEDAR_ENTROPY_FUSED=0
run "round-3 back end, coherent host memory" CEDAR_ENTROPY_FUSED=0 CEDAR_HOST_COHERENT=1
