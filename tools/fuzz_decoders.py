#!/usr/bin/env python3
"""Differential fuzz of the two decoders that check the encoders: oracle/h264o_dec.c (C, same author as the
encoder twin) and tests/indep/h264_refdec.py (Python, own code and tables). Valid streams of the oracle are
damaged -- single bit flips, byte overwrites, truncations, in the slice data and in the parameter sets -- and
both decoders read every damaged stream. They must AGREE: both refuse it, or both accept it and rebuild the
same pictures. A stream only one of them accepts, or pictures that differ, is a finding (one of the two reads
the standard differently, or trusts a field the other checks).
    python tools/fuzz_decoders.py [--seed N] [--cases N]      exits 1 on the first finding
tests/test_independent_decoder.py::test_decoders_agree_on_damaged_streams runs a short campaign on the CPU."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import oracle_lib as ol                      # noqa: E402
from tests.indep import h264_refdec as rd               # noqa: E402

# w, h, qp, keyint, frames, pattern, slices, extra
STREAMS = [
    (64, 48, 30, 3, 4, "B", 1, {}), (48, 48, 24, 25, 3, "C", 1, {}), (64, 64, 36, 2, 3, "B", 2, {}),
    (96, 64, 28, 25, 3, "B", 1, dict(subpel=2)), (64, 48, 30, 25, 3, "B", 1, dict(intra_refresh=3)),
    (64, 48, 26, 25, 3, "B", 1, dict(roi=[(0, 0, 2, 2, -6)])),
]


def make_streams():
    out = []
    for w, h, qp, keyint, frames, pat, slices, extra in STREAMS:
        oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=keyint, slice_count=slices, **extra))
        out.append([oe.encode(*ol.gen_frame(pat, w, h, f)) for f in range(frames)])
    return out


def read_c(aus):
    """(accepted, pictures) from the C decoder; a refused access unit ends the stream"""
    dec = ol.OracleDecoder()
    pics = []
    try:
        for au in aus:
            n = dec.decode(au)
            if n:
                y, uv = dec.picture()
                pics.append(y.tobytes() + uv.tobytes())
    except ValueError as ex:
        return False, pics, str(ex)
    return True, pics, ""


def read_py(aus):
    dec = rd.Decoder()
    pics = []
    try:
        for au in aus:
            for p in dec.decode(au):
                pics.append(p.luma_bytes() + p.nv12_chroma_bytes())
    except rd.SyntaxError264 as ex:
        return False, pics, str(ex)
    return True, pics, ""


def damage(rng, aus):
    aus = [bytearray(a) for a in aus]
    k = int(rng.integers(len(aus)))
    a = aus[k]
    kind = int(rng.integers(4))
    lo = 5 if rng.random() < 0.85 else 0   # mostly past the first start code + NAL header
    if kind == 0:
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(lo, len(a)))
            a[pos] ^= 1 << int(rng.integers(8))
    elif kind == 1:
        pos = int(rng.integers(lo, len(a)))
        a[pos] = int(rng.integers(256))
    elif kind == 2:
        del a[int(rng.integers(max(lo, len(a) // 2), len(a))):]
    else:
        pos = int(rng.integers(lo, len(a)))
        a[pos:pos] = bytes(int(x) for x in rng.integers(0, 256, int(rng.integers(1, 4))))
    return [bytes(x) for x in aus], (k, kind)


def campaign(seed, cases, verbose=False):
    rng = np.random.default_rng(seed)
    streams = make_streams()
    stats = {"both_refuse": 0, "both_accept_same": 0, "undamaged": 0}
    findings = []
    for i in range(cases):
        si = int(rng.integers(len(streams)))
        aus, what = damage(rng, streams[si])
        okc, pc, errc = read_c(aus)
        okp, pp, errp = read_py(aus)
        if ((not okc and "unsupported" in errc) or (not okp and "outside this decoder" in errp)) and okc != okp:
            stats["out_of_scope"] = stats.get("out_of_scope", 0) + 1   # a feature one of the two does not implement: no verdict
        elif okc != okp:
            findings.append((i, si, what, "only the %s decoder accepts it (%s)" % ("C" if okc else "Python", errp if okc else errc)))
        elif okc and pc != pp:
            findings.append((i, si, what, "both accept it, pictures differ"))
        elif okc:
            stats["both_accept_same"] += 1
        else:
            # up to the access unit that was refused both must have rebuilt the same pictures
            n = min(len(pc), len(pp))
            if pc[:n] != pp[:n]:
                findings.append((i, si, what, "refused by both, but the pictures before differ"))
            stats["both_refuse"] += 1
        if verbose and findings and findings[-1][0] == i:
            print("case %d stream %d damage %s: %s" % findings[-1], flush=True)
    return stats, findings


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cases", type=int, default=400)
    a = ap.parse_args()
    stats, findings = campaign(a.seed, a.cases, verbose=True)
    print(stats, "findings:", len(findings))
    sys.exit(1 if findings else 0)


if __name__ == "__main__":
    main()
