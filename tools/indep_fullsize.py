#!/usr/bin/env python3
"""Writes tests/golden/indep_fullsize.json: the first GOP (25 access units) of BASELINE configs[1], [2] and
[3] -- 854x480 QP 30, 1920x1080 QP 24, 3840x2160 in 4 slices QP 24, pattern-B -- as the C oracle emits them,
read by the INDEPENDENT Python decoder (tests/indep/h264_refdec.py: own code, own tables). Every picture it
rebuilds must equal the oracle's in-loop reconstruction; the stream must stay within the level its SPS names
(Decoder.level_violations). Committed per access unit: the SHA-1 of its bytes; per case: the SHA-1 of all
bytes and of all pictures. The GPU suite then only has to hash the HIP path's bytes against these
(tests/test_gpu_parity.py::test_hip_streams_equal_the_fullsize_golden_vectors): at BASELINE's full sizes
the bytes are pinned to a reading of the standard that shares nothing with oracle/ or csrc/.
Round 4: BASELINE configs[4] is 8 concurrent 1080p streams; stream 0 is configs[2]'s, the other seven differ in
content (pattern-B with seed offset s << 8, what bench.py feeds stream s): their first 3 access units each.
Run from the repo root: python tools/indep_fullsize.py   (one process per case; the 4K case takes ~10 minutes;
`--only-missing` keeps the cases the JSON already holds)"""
import hashlib
import json
import multiprocessing
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# name, width, height, qp, keyint, frames, slices, fps (VUI timing: MaxMBPS is checked at it)
CASES = [
    ("configs[1]", 854, 480, 30, 25, 25, 1, 30),
    ("configs[2]", 1920, 1080, 24, 25, 25, 1, 30),
    ("configs[3]", 3840, 2160, 24, 25, 25, 4, 30),
] + [("configs[4] stream %d" % s, 1920, 1080, 24, 25, 3, 1, 30, s << 8) for s in range(1, 8)]


def run_case(case):
    from tests import oracle_lib as ol
    from tests.indep import h264_refdec as rd
    name, w, h, qp, keyint, frames, slices, fps = case[:8]
    seed = case[8] if len(case) > 8 else 0
    level = 51 if ((w + 15) // 16) * ((h + 15) // 16) > 8192 else 41
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=keyint, slice_count=slices, level_idc=level, fps_num=fps, fps_den=1))
    dec = rd.Decoder()
    stream, pics, aus = hashlib.sha1(), hashlib.sha1(), []
    t0 = time.time()
    for f in range(frames):
        au = oe.encode(*ol.gen_frame("B", w, h, f, seed))
        stream.update(au)
        aus.append(hashlib.sha1(au).hexdigest())
        out = dec.decode(au)
        assert len(out) == 1, "one access unit, one picture"
        ly, lc = out[0].luma_bytes(), out[0].nv12_chroma_bytes()
        ry, ruv = oe.recon()
        if ly != ry.tobytes() or lc != ruv.tobytes():
            raise SystemExit("%s frame %d: the independent decoder and the oracle's reconstruction differ" % (name, f))
        pics.update(ly)
        pics.update(lc)
        print("%s frame %d: %d bytes, %.0f s" % (name, f, len(au), time.time() - t0), flush=True)
    v = dec.level_violations()
    if v:
        raise SystemExit("%s: %s" % (name, v))
    return {"case": list(case), "level_idc": level, "stream_sha1": stream.hexdigest(), "pictures_sha1": pics.hexdigest(),
            "au_sha1": aus, "max_mb_bits": dec.max_mb_bits, "max_abs_mv_quarter": dec.max_abs_mv}


def main():
    path = os.path.join(ROOT, "tests", "golden", "indep_fullsize.json")
    have = {}
    if "--only-missing" in sys.argv and os.path.exists(path):
        have = {c["case"][0]: c for c in json.load(open(path))["cases"]}
    todo = [c for c in CASES if c[0] not in have]
    with multiprocessing.Pool(min(len(todo), 7) or 1) as pool:
        new = {c["case"][0]: c for c in pool.map(run_case, todo)}
    out = [have.get(c[0]) or new[c[0]] for c in CASES]
    json.dump({"_made_by": "tools/indep_fullsize.py", "cases": out}, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
