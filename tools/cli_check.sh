#!/bin/bash
# Runs ON THE GPU BOX: the GPU CLI against the scalar CLI on the same clip (development tool).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python3 - <<'PY'
import sys, numpy as np
sys.path.insert(0, ".")
from tests import oracle_lib as ol
with open("/tmp/clip.nv12", "wb") as f:
    for i in range(23):
        y, uv = ol.gen_frame("B", 854, 480, i)
        f.write(y.tobytes()); f.write(uv.tobytes())
PY
export H264ENC_QP=30 H264ENC_KEYINT=10
cat /tmp/clip.nv12 | cedarx_h264_encoder_amd/lib/h264enc - 854 480 /tmp/gpu.264 > /dev/null
cat /tmp/clip.nv12 | oracle/build/h264enc_scalar - 854 480 /tmp/cpu.264 > /dev/null
cmp /tmp/gpu.264 /tmp/cpu.264 && echo "CLI pipelined: identical ($(stat -c %s /tmp/gpu.264) bytes)"
cat /tmp/clip.nv12 | H264ENC_PIPELINE=0 cedarx_h264_encoder_amd/lib/h264enc - 854 480 /tmp/gpu0.264 > /dev/null
cmp /tmp/gpu0.264 /tmp/cpu.264 && echo "CLI serial: identical"
oracle/build/h264dec /tmp/gpu.264 /tmp/dec.nv12 /tmp/clip.nv12 | tail -1
