#!/bin/bash
# Runs ON THE GPU BOX: one 1080p stream: analysis streams x deblock streams x diagonal stripes per P picture (development tool).
run() { echo "analysis_streams=$1 deblock_streams=$2 stripes=$3"; CEDAR_ANALYSIS_STREAMS=$1 CEDAR_DEBLOCK_STREAMS=$2 CEDAR_INTER_BANDS=$3 timeout -k 10 120 python3 tools/tput.py 1 2>/dev/null | tail -1; }
run 1 2 5 && run 2 2 5 && run 2 2 8 && run 2 3 5 && run 2 3 8
