#!/bin/bash
# Runs ON THE GPU BOX: per-stage times for several (deblock split, intra split) pairs (development tool).
run() { echo "S=$1 split_deblock=$2 split_intra=$3"; CEDAR_WF_SPLIT=$2 CEDAR_WF_SPLIT_INTRA=$3 timeout -k 10 120 python3 tools/stage_times.py $1 2>/dev/null | tail -1; }
run 1 2 4 && run 1 3 6 && run 1 4 8 && run 1 6 8 && run 1 8 8 &&
run 8 2 3 && run 8 3 4 && run 8 4 6 && run 8 8 8 &&
run 32 1 2 && run 32 2 3 && run 32 3 4 && run 32 4 6
