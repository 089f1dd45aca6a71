"""Long pipelined runs with the quality features on, every access unit against the oracle (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import cedarx_h264_encoder_amd as ce
from tests import oracle_lib as ol
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
cases = [
    (640, 368, 26, 25, dict(roi=((4, 3, 12, 8, -7), (0, 0, 40, 23, 3)), mad_th=(1, 3, 6, 12), mad_dqp=(-3, -1, 0, 2, 4), intra_refresh=23, subpel=2)),
    (854, 480, 30, 12, dict(intra_refresh=5, subpel=1, slice_count=3)),
    (320, 240, 24, 7, dict(roi=((2, 2, 6, 6, -12),), bitrate=900000, aud=1, repeat_headers=1)),
    (416, 240, 33, 25, dict(mad_th=(0, 2, 4, 8), mad_dqp=(5, 2, 0, -2, -5), strict_mb_bits=2)),
]
t00 = time.time()
for w, h, qp, K, kw in cases:
    okw = {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()}
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=K, **okw))
    t0 = time.time()
    with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=K, **kw) as enc:
        depth, q, got = enc.pipeline_depth, 0, 0
        pend = []
        def take():
            global got
            enc.collect()
            f = pend.pop(0)
            ref = oe.encode(*ol.gen_frame("BC"[(f // 40) & 1], w, h, f))
            if enc.output(0) != ref:
                sys.exit("MISMATCH %dx%d %s frame %d" % (w, h, kw, f))
            got += 1
        for f in range(n):
            if q == depth:
                take(); q -= 1
            y, uv = ol.gen_frame("BC"[(f // 40) & 1], w, h, f)
            enc.set_input(y, uv)
            enc.submit(); q += 1; pend.append(f)
        while q:
            take(); q -= 1
    print("%dx%d %s: %d frames bit-exact in %.0f s" % (w, h, kw, got, time.time() - t0), flush=True)
print("soak done in %.0f s" % (time.time() - t00))
