#!/bin/bash
# Runs ON THE GPU BOX: frames per second of the command line itself (file in tmpfs -> stdin pipe ->
# pinned buffer -> H2D -> encode -> D2H -> file), pipelined and serial (development tool).
set -e
cd ${GRAFT_REPO_ROOT:-.}
N=${1:-300}
W=${2:-1920}
H=${3:-1080}
python3 - <<PY
import sys
sys.path.insert(0, ".")
import cedarx_h264_encoder_amd as ce
with open("/dev/shm/cli_rate.nv12", "wb") as f:
    for i in range($N):
        y, uv = ce.gen_pattern("B", $W, $H, i % 50, 0)
        f.write(y.tobytes()); f.write(uv.tobytes())
PY
for mode in 1 0; do
    s=$(date +%s.%N)
    H264ENC_PIPELINE=$mode ./cedarx_h264_encoder_amd/lib/h264enc - $W $H /dev/shm/cli_rate_$mode.264 < /dev/shm/cli_rate.nv12 > /dev/null
    e=$(date +%s.%N)
    python3 -c "print('H264ENC_PIPELINE=$mode: %d frames ${W}x${H} in %.3f s = %.0f frames/s' % ($N, $e - $s, $N / ($e - $s)))"
done
cmp /dev/shm/cli_rate_1.264 /dev/shm/cli_rate_0.264 && echo "same file"
rm -f /dev/shm/cli_rate.nv12 /dev/shm/cli_rate_1.264 /dev/shm/cli_rate_0.264
