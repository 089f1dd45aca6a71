#!/usr/bin/env python3
"""Development: the session of the round's one unreproduced fuzz mismatch (2 streams x 3 GOP lanes, 32x16, QP 1, keyint 5,
half-sample vectors, cyclic refresh 7, MAD classes) again and again in one process, with varying content and length,
against the oracle. Arguments: sessions (300). Environment: CEDAR_* as usual."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import cedarx_h264_encoder_amd as ce
from tests import oracle_lib as ol
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.RandomState(7)
kw = dict(aud=0, fps_num=25, fps_den=1001, subpel=1, intra_refresh=7, mad_th=[1, 5, 10, 12], mad_dqp=[0, -9, -10, 3, 3])
for k in os.environ.get("STRESS_DROP", "").split(","):   # options to leave out (bisecting)
    if k == "mad":
        kw.pop("mad_th"); kw.pop("mad_dqp")
    elif k:
        kw.pop(k)
QP = int(os.environ.get("STRESS_QP", "1"))
bad, t0 = 0, time.time()
for it in range(N):
    w, h = int(rng.choice([32, 32, 48, 64])), int(rng.choice([16, 16, 32]))
    S, G, K = int(os.environ.get("STRESS_S", "2")), int(os.environ.get("STRESS_G", "3")), 5
    n = int(rng.randint(6, 40))
    off = int(rng.randint(0, 1000))
    frames = [[ol.gen_frame("C", w, h, f, off + (s << 8)) for s in range(S)] for f in range(n)]
    oes = [ol.OracleEncoder(ol.default_config(w, h, qp=QP, keyint=K, **kw)) for _ in range(S)]
    want = [[oes[s].encode(*frames[f][s]) for s in range(S)] for f in range(n)]
    with ce.CedarEncoder(width=w, height=h, qp=QP, keyframe_interval=K, n_streams=S, gop_parallel=G if G > 1 else 0, **kw) as enc:
        depth, got, q = enc.pipeline_depth, [], 0
        for fr in frames:
            if q == depth:
                enc.collect(); got.append([enc.output(s) for s in range(S)]); q -= 1
            for s in range(S):
                enc.set_input(fr[s][0], fr[s][1], stream=s)
            enc.submit(); q += 1
        enc.flush()
        while q:
            enc.collect(); got.append([enc.output(s) for s in range(S)]); q -= 1
    if got != want:
        bad += 1
        where = [(f, s, len(got[f][s]), len(want[f][s])) for f in range(n) for s in range(S) if got[f][s] != want[f][s]]
        print("MISMATCH session %d: %dx%d n %d offset %d: %s of %d" % (it, w, h, n, off, where[:4], len(where)), flush=True)
        if bad <= 2:   # what kind of difference: the first differing access unit of both sides, and whether ours decodes
            f0, s0 = where[0][0], where[0][1]
            g_, w_ = got[f0][s0], want[f0][s0]
            k = next((i for i in range(min(len(g_), len(w_))) if g_[i] != w_[i]), min(len(g_), len(w_)))
            print("  first differing byte %d; ours %s ... oracle %s" % (k, g_[max(0, k - 8):k + 24].hex(), w_[max(0, k - 8):k + 24].hex()))
            try:
                dec = ol.OracleDecoder()
                for f in range(f0 + 1):
                    r = dec.decode(got[f][s0])
                print("  our access unit decodes:", r)
            except Exception as ex:  # noqa: BLE001
                print("  our access unit does not decode:", ex)
print("%d of %d sessions differ, %.0f s" % (bad, N, time.time() - t0), {k: v for k, v in os.environ.items() if k.startswith(("CEDAR_", "STRESS_"))})
