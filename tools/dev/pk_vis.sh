#!/bin/bash
# GPU box: the pipelined tests that showed same-length access units with other bytes, under k_pack_out's delivery modes
cd $GRAFT_REPO_ROOT
for mode in 0 2 1; do
  for rep in 1 2 3; do
    CEDAR_PK_FENCE=$mode timeout -k 10 280 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "long_pipelined or frame_pipelining_on_off or gop_parallel_session or fuzz_sweep" 2>&1 | tail -1 | sed "s/^/mode $mode rep $rep: /"
  done
done
