import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cedarx_h264_encoder_amd as ce
enc = ce.CedarEncoder(width=1920, height=1080, qp=24, keyframe_interval=25)
for f in range(4):
    y, uv = ce.gen_pattern("B", 1920, 1080, f, 0)
    enc.encode_frame(y, uv)
enc.close()
