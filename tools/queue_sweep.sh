#!/bin/bash
# Runs ON THE GPU BOX: throughput of one 1080p stream for several (hardware queues, analysis
# streams, deblock streams) choices (development tool).
run() { echo "queues=$1 analysis=$2 deblock=$3"; GPU_MAX_HW_QUEUES=$1 CEDAR_ANALYSIS_STREAMS=$2 CEDAR_DEBLOCK_STREAMS=$3 timeout -k 10 120 python3 tools/tput.py 1 2>/dev/null | tail -1; }
run 8 1 2 && run 8 1 3 && run 8 2 2 && run 8 2 3 && run 8 3 3 && run 5 1 2 && run 6 1 2 && run 6 2 2 && run 6 1 3
