#!/usr/bin/env python3
"""Writes tests/golden/indep_decode.json: for a fixed list of small encodes, the SHA-1 of the
Annex-B stream the C oracle emits and the SHA-1 of the pictures that the INDEPENDENT Python
decoder (tests/indep/h264_refdec.py: own code, own tables) makes of it. The vectors pin the
oracle (and through it the HIP path, which must emit the same bytes) to a reading of the
standard that shares nothing with oracle/ -- see DESIGN.md section 2.
Run from the repo root: python tools/gen_indep_golden.py"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import oracle_lib as ol                      # noqa: E402
from tests.indep import h264_refdec as rd               # noqa: E402

# w, h, qp, keyint, frames, pattern, slices [, "option=value,..." of the oracle's configuration]
CASES = [
    (64, 48, 30, 25, 4, "B", 1), (64, 48, 24, 25, 3, "C", 1), (64, 48, 30, 25, 3, "A", 1), (70, 50, 24, 25, 3, "B", 1),
    (64, 48, 5, 3, 4, "C", 1, "pcm=-1"), (64, 48, 47, 3, 4, "B", 1), (16, 16, 24, 2, 3, "C", 1), (48, 80, 24, 4, 5, "B", 5),
    (96, 64, 36, 4, 5, "B", 2), (64, 48, 1, 2, 2, "C", 1, "pcm=-1"), (128, 96, 40, 25, 3, "B", 1), (96, 112, 18, 25, 3, "C", 1),
    (16, 96, 24, 3, 6, "B", 1), (320, 240, 30, 25, 3, "B", 1), (320, 240, 24, 1, 2, "B", 4),
    (96, 64, 28, 25, 4, "B", 1, "subpel=2"), (96, 64, 28, 12, 6, "B", 1, "intra_refresh=3"),
    (112, 80, 30, 25, 5, "B", 2, "subpel=2,intra_refresh=5"), (48, 48, 20, 30, 9, "C", 1, "intra_refresh=7"),
    (160, 96, 28, 25, 4, "B", 2, "subpel=2,roi=0:0:4:3:-6/2:1:6:4:10/9:5:1:1:-20"), (96, 64, 36, 6, 7, "C", 1, "roi=1:1:3:2:-12"),
    (128, 96, 30, 25, 5, "B", 1, "mad_th=2:5:10:20,mad_dqp=-4:-2:0:2:4"), (112, 80, 27, 8, 6, "B", 2, "subpel=2,intra_refresh=4,roi=0:0:3:3:8,mad_th=1:3:6:12,mad_dqp=6:3:0:-3:-6"),
    (64, 48, 26, 3, 5, "C", 1, "roi=0:0:1:1:24/1:0:1:1:-24/2:1:1:2:24"),   # mb_qp_delta of +-48 goes round modulo 52
    # I_PCM: noise at a low QP (every macroblock), at QP 18 (most, some not: the neighbours of both kinds), in slices
    (64, 48, 5, 3, 4, "C", 1, "pcm=0"), (96, 64, 18, 4, 5, "C", 1, "pcm=0"), (48, 80, 12, 25, 4, "C", 3, "pcm=0"), (64, 48, 1, 2, 3, "C", 1, "subpel=2,pcm=0"),
]


def case_options(case):
    """the optional 8th element as keyword arguments (the names are h264o_config's and CedarEncoder's)"""
    opts = {}
    for k, v in (kv.split("=") for kv in case[7].split(",")) if len(case) > 7 else ():
        if k == "roi":                    # roi=x:y:w:h:dqp/...
            opts[k] = [tuple(int(t) for t in r.split(":")) for r in v.split("/")]
        elif k in ("mad_th", "mad_dqp"):  # mad_th=t0:t1:t2:t3
            opts[k] = [int(t) for t in v.split(":")]
        else:
            opts[k] = int(v)
    return opts


def encode_and_decode(case):
    w, h, qp, keyint, frames, pattern, slices = case[:7]
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=keyint, slice_count=slices, **case_options(case)))
    dec = rd.Decoder()
    stream, pics, same = hashlib.sha1(), hashlib.sha1(), True
    for f in range(frames):
        y, uv = ol.gen_frame(pattern, w, h, f)
        au = oe.encode(y, uv)
        stream.update(au)
        out = dec.decode(au)
        assert len(out) == 1, "one access unit, one picture"
        ly, lc = out[0].luma_bytes(), out[0].nv12_chroma_bytes()
        pics.update(ly)
        pics.update(lc)
        ry, ruv = oe.recon()
        same = same and ly == ry.tobytes() and lc == ruv.tobytes()
    return stream.hexdigest(), pics.hexdigest(), same, dec.cover, dec.level_violations()


def main():
    out = []
    for case in CASES:
        s, p, same, _, _ = encode_and_decode(case)
        if not same:
            raise SystemExit("case %s: the independent decoder and the oracle's reconstruction differ" % (case,))
        out.append({"case": list(case), "stream_sha1": s, "pictures_sha1": p})
    path = os.path.join(ROOT, "tests", "golden", "indep_decode.json")
    json.dump({"_made_by": "tools/gen_indep_golden.py", "cases": out}, open(path, "w"), indent=1)
    print("wrote", path, len(out), "cases")


if __name__ == "__main__":
    main()
