#!/bin/bash
# Runs ON THE GPU BOX: one 1080p stream, deblock streams x diagonal stripes per P picture (development tool).
run() { echo "deblock_streams=$1 stripes=$2"; CEDAR_DEBLOCK_STREAMS=$1 CEDAR_INTER_BANDS=$2 timeout -k 10 120 python3 tools/tput.py 1 2>/dev/null | tail -1; }
run 2 5 && run 2 8 && run 2 12 && run 3 5 && run 3 8 && run 3 12 && run 3 16
