import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import cedarx_h264_encoder_amd as ce
from tests import oracle_lib as ol
import torch
w,h,K,G,n = 1920,1080,4,3,14
frames=[ol.gen_frame("B",w,h,f) for f in range(n)]
dev=[(torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda()) for y,uv in frames]
torch.cuda.synchronize()
with ce.CedarEncoder(width=w,height=h,qp=27,keyframe_interval=K,gop_parallel=G, gop_low_latency=int(os.environ.get("LOW","0"))) as enc:
    for i in range(n):
        enc.set_device_input(0, dev[i][0].data_ptr(), dev[i][1].data_ptr())
        enc.submit()
    enc.flush()
    for i in range(n):
        try:
            enc.collect(); print("frame", i, "ok", len(enc.output(0)))
        except Exception as ex:
            print("frame", i, "ERR", ex)
