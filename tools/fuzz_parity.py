#!/usr/bin/env python3
"""One-off wide parity sweep against the oracle: random geometries, QPs, GOP lengths, slice and
stream counts, header options, frames kept in flight (development tool; tests/ holds the fixed cases)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cedarx_h264_encoder_amd as ce
from tests import oracle_lib as ol
seed, n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 100
first_case = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rng = np.random.RandomState(seed)
t0 = time.time()
for case in range(n_cases):
    w = max(16, int(rng.randint(1, 40)) * 16 - int(rng.choice([0, 0, 2, 6, 10, 14])))
    h = max(16, int(rng.randint(1, 24)) * 16 - int(rng.choice([0, 0, 2, 4, 8, 12])))
    if os.environ.get("FUZZ_NARROW") and case % 2 == 0:   # every other case: one to three macroblocks wide or high
        if case % 4 == 0:
            w = int(rng.choice([16, 18, 32, 34, 48]))
        else:
            h = int(rng.choice([16, 18, 32, 34, 48]))
    if os.environ.get("FUZZ_BIG"):   # 640x360 .. 1920x1088: many bands per workgroup, the real split factors
        w = int(rng.randint(40, 121)) * 16 - int(rng.choice([0, 0, 2, 6, 10, 14]))
        h = int(rng.randint(22, 69)) * 16 - int(rng.choice([0, 0, 2, 4, 8, 12]))
    mb_h = (h + 15) // 16
    qp = int(rng.choice([1, 8, 17, 24, 30, 36, 41, 47]))
    keyint = int(rng.choice([1, 2, 3, 5, 25]))
    slices = int(rng.randint(1, min(4, mb_h) + 1))
    S = int(rng.choice([1, 1, 1, 2, 3, 9]))
    pat = str(rng.choice(["A", "B", "B", "C"]))
    kw = dict(aud=int(rng.randint(0, 2)), repeat_headers=int(rng.randint(0, 2)))
    if rng.randint(0, 2):
        kw.update(fps_num=int(rng.choice([24, 25, 30000])), fps_den=int(rng.choice([1, 1001])))
    if os.environ.get("FUZZ_WIDE"):   # more of the configuration space: many streams, search ranges, chroma QP offsets
        S = int(rng.choice([1, 2, 8, 9, 17, 33]))
        if S > 9:
            w, h = min(w, 160), min(h, 96)
        sr = int(rng.choice([0, 0, -1, 1, 4, 9]))
        coff = int(rng.choice([4, 4, -12, 0, 12]))
        kw.update(search_range=sr, chroma_qp_offset_minus4=coff - 4)
    n = int(rng.randint(3, 14))
    if os.environ.get("FUZZ_BIG"):
        n, S = min(n, 5), min(S, 2)
    depth_draw = int(rng.randint(1, 17))
    G = 0
    if os.environ.get("FUZZ_R2"):   # round 2's options: sub-sample vectors, the macroblock bit cap, rate control, GOP lanes
        kw.update(subpel=int(rng.choice([0, 1, 2, 2])))
        pick = int(rng.randint(0, 4))
        if S == 1 and pick == 1:
            kw.update(strict_mb_bits=2)
        elif S <= 9 and pick == 2:   # (rate control runs per stream)
            n = int(rng.randint(19, 30))
            kw.update(bitrate=int(rng.choice([1, 4, 20])) * w * h)
        elif pick == 3 and S <= 3:
            G = int(rng.randint(2, 5))
            n = int(rng.randint(3, 3 * keyint * G + 3))
            kw.pop("repeat_headers", None)
    if os.environ.get("FUZZ_REFRESH"):   # cyclic intra refresh on top of whatever the case drew
        kw.update(intra_refresh=int(rng.choice([3, 3, 4, 7, 50])))
        if "subpel" not in kw:
            kw.update(subpel=int(rng.choice([0, 2])))
    if os.environ.get("FUZZ_ROI"):   # regions of interest on top of whatever the case drew
        mbw, mbh = (w + 15) // 16, (h + 15) // 16
        kw.update(roi=[(int(rng.randint(0, mbw)), int(rng.randint(0, mbh)), int(rng.randint(1, mbw + 1)), int(rng.randint(1, mbh + 1)),
                        int(rng.randint(-24, 25))) for _ in range(int(rng.randint(1, 5)))])
    if os.environ.get("FUZZ_MAD") and rng.randint(0, 3):   # MAD classes on top
        kw.update(mad_th=sorted(int(x) for x in rng.randint(0, 40, 4)), mad_dqp=[int(x) for x in rng.randint(-12, 13, 5)])
    if case < first_case:
        continue
    print("case %d: %dx%d qp %d keyint %d slices %d streams %d pattern %s %s n %d depth<=%d" % (case, w, h, qp, keyint, slices, S, pat, kw, n, depth_draw), flush=True)
    frames = [[ol.gen_frame(pat, w, h, f, s << 8) for s in range(S)] for f in range(n)]
    okw = dict(kw)
    if "search_range" in okw:
        sr_ = okw.pop("search_range")
        okw["search_range"] = 16 if sr_ == 0 else max(sr_, 0)
        okw["chroma_qp_offset"] = okw.pop("chroma_qp_offset_minus4") + 4
    if S > 1 and okw.get("bitrate", 0) > 0:
        okw["pcm"] = -1   # rate control over several lock-step streams: such a session takes no I_PCM (cedar_h264enc.h, pcm)
    oes = [ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=keyint, slice_count=slices, **okw)) for _ in range(S)]
    want = [[oes[s].encode(*frames[f][s]) for s in range(S)] for f in range(n)]
    if os.environ.get("FUZZ_WIDE") and slices >= 2 and S == 1 and case % 3 == 0 and not ({"strict_mb_bits", "bitrate"} & set(kw)):
        # the slices of one picture on two sessions ("two GPUs"): the concatenation is the whole
        a = int(rng.randint(1, slices))
        parts = [ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint, slice_count=slices,
                                 slice_first=f0, slice_num=fn, **kw) for f0, fn in ((0, a), (a, slices - a))]
        got = []
        for fr in frames:
            for pe in parts:
                pe.set_input(fr[0][0], fr[0][1])
                pe.submit()
            au = b""
            for pe in parts:
                pe.collect()
                au += pe.output(0)
            got.append([au])
        for pe in parts:
            pe.close()
        if got != want:
            sys.exit("MISMATCH (sharded slices %d+%d) seed %d case %d: %dx%d qp %d keyint %d slices %d pattern %s %s" % (
                a, slices - a, seed, case, w, h, qp, keyint, slices, pat, kw))
        continue
    with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint, slice_count=slices, n_streams=S, gop_parallel=G, **kw) as enc:
        depth, got, q = min(enc.pipeline_depth, depth_draw), [], 0
        if G:
            depth = enc.pipeline_depth
        for fr in frames:
            if q == depth:
                enc.collect(); got.append([enc.output(s) for s in range(S)]); q -= 1
            for s in range(S):
                enc.set_input(fr[s][0], fr[s][1], stream=s)
            enc.submit(); q += 1
        enc.flush()
        while q:
            enc.collect(); got.append([enc.output(s) for s in range(S)]); q -= 1
        st = enc.stats()
    if got != want:
        # (which access unit, and under which schedule: an intermittent mismatch should leave something to go on)
        where = [(f, s, len(got[f][s]), len(want[f][s])) for f in range(min(len(got), len(want))) for s in range(S) if got[f][s] != want[f][s]]
        sys.exit("MISMATCH seed %d case %d: %dx%d qp %d keyint %d slices %d streams %d lanes %d pattern %s %s depth %d; %d of %d frames collected, "
                 "first differing (frame, stream, bytes, oracle's bytes) %s of %d; deblock_rows %s pipelined %s look-ahead %s" % (
            seed, case, w, h, qp, keyint, slices, S, G, pat, kw, depth, len(got), len(want), where[:3], len(where),
            st["deblock_rows"], st["frames_pipelined"], st["idr_lookahead"]))
print("seed %d: %d cases bit-exact in %.0f s" % (seed, n_cases, time.time() - t0))
