"""The encoders' streams read by a decoder that shares nothing with them.

tests/indep/h264_refdec.py is a plain-Python H.264 baseline decoder written from the text of the
standard, with tables typed in for that file (the reference holds no vectors for the macroblock
stages and no box here has a third-party decoder: DESIGN.md section 2). These tests
  * check its VLC tables structurally (prefix-free; Kraft sums exactly what the standard's tables
    have: complete but for the all-zero code word it keeps out of coeff_token / total_zeros /
    run_before);
  * decode what the C oracle emits for fifteen small cases (I and P pictures, noise at QP 1..47,
    five slices with deblocking idc 2, pictures one macroblock wide, cropped sizes) and compare every
    picture with the oracle's in-loop reconstruction, with the oracle's own C decoder, and with the
    committed vectors of tests/golden/indep_decode.json (made by tools/gen_indep_golden.py);
  * insist that those streams exercised every Intra4x4 / Intra16x16 / chroma predictor, every
    boundary strength on both planes and the strong filter, every total_zeros and run_before row,
    all five coeff_token tables, the level escape codes and every suffixLength, every motion vector
    predictor case and both P_Skip rules.
CPU only; the GPU suite decodes the HIP path's bytes with the same decoder."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from tests import oracle_lib as ol                      # noqa: E402
from tests.indep import h264_refdec as rd               # noqa: E402
import gen_indep_golden as gen                          # noqa: E402


def test_vlc_tables_are_prefix_free_and_as_complete_as_the_standards():
    deficit = {  # 1 - Kraft sum: the one code word (all zeros, at the longest length) each table leaves out
        "coeff_token 0<=nC<2": 2.0 ** -15, "coeff_token 2<=nC<4": 2.0 ** -13, "coeff_token 4<=nC<8": 2.0 ** -10,
        "coeff_token 8<=nC": 2.0 / 64, "total_zeros 4x4 tzVlcIndex 1": 2.0 ** -9, "run_before zerosLeft 7": 2.0 ** -11,
    }
    tabs = rd.vlc_tables()
    assert len(tabs) == 5 + 15 + 3 + 7
    for name, tab in tabs.items():
        codes = [format(v, "0%db" % ln) for ln, v in tab.values()]
        assert all(v < (1 << ln) for ln, v in tab.values()), name
        assert len(set(codes)) == len(codes), name
        assert not any(a != b and b.startswith(a) for a in codes for b in codes), name
        kraft = sum(2.0 ** -len(c) for c in codes)
        assert abs(1.0 - kraft - deficit.get(name, 0.0)) < 1e-12, (name, kraft)
    assert sorted(a for a, _ in rd.CBP_ME) == list(range(48)) and sorted(b for _, b in rd.CBP_ME) == list(range(48))


def test_textbook_cavlc_block_is_read_back():
    # zig-zag [0,3,0,1,-1,-1,0,1,0...], 0 <= nC < 2 -> 000010001110010111101101
    bits = "000010001110010111101101" + "1"
    data = int(bits.ljust(32, "0"), 2).to_bytes(4, "big")
    d = rd.Decoder()
    coef, total = d._residual_block(rd.Bits(data), 0, 16)
    assert total == 5 and coef == [0, 3, 0, 1, -1, -1, 0, 1] + [0] * 8


@pytest.fixture(scope="module")
def decoded():
    return [(case, gen.encode_and_decode(case)) for case in gen.CASES]


def test_oracle_streams_decode_to_the_oracle_reconstruction(decoded):
    golden = {tuple(c["case"]): c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "indep_decode.json")))["cases"]}
    assert len(golden) == len(gen.CASES)
    for case, (stream_sha, pics_sha, same, _, _) in decoded:
        assert same, "case %s: independent decode != oracle reconstruction" % (case,)
        assert golden[case]["stream_sha1"] == stream_sha, "case %s: the oracle's bytes changed" % (case,)
        assert golden[case]["pictures_sha1"] == pics_sha, case


def test_the_streams_reach_every_part_of_the_standard_the_encoders_use(decoded):
    from collections import Counter
    cover = Counter()
    for _, (_, _, _, c, _) in decoded:
        cover.update(c)
    want = (["i4_mode_%d" % i for i in range(9)] + ["i16_mode_%d" % i for i in range(4)] + ["chroma_mode_%d" % i for i in range(4)] +
            ["deblock_luma_bs%d" % i for i in range(1, 5)] + ["deblock_chroma_bs%d" % i for i in range(1, 5)] +
            ["deblock_luma_bs4_strong", "deblock_idc_0", "deblock_idc_2"] +
            ["total_zeros_row_%d" % i for i in range(1, 16)] + ["total_zeros_cdc_row_%d" % i for i in range(1, 4)] +
            ["run_before_row_%d" % i for i in range(1, 8)] + ["suffix_length_%d" % i for i in range(7)] +
            ["level_prefix_14", "level_prefix_15", "mvp_median", "mvp_only_A", "mvp_single_match", "mvp_D_for_C",
             "skip_mv_zero_rule", "skip_mv_predicted", "mb_P_Skip", "mb_I4x4", "mb_I16x16", "mb_P_L0_16x16",
             "mb_skip_run_0", "mb_skip_run_1", "mb_skip_run_n", "i4_top_right_substituted", "i4_mode_predicted", "i4_mode_rem",
             "mvd_zero", "mvd_nonzero", "mc_chroma_frac_0_0", "mc_chroma_frac_4_0", "mc_chroma_frac_0_4", "mc_chroma_frac_4_4",
             "constrained_intra_neighbour_hidden", "mb_I_PCM"] +
            ["coeff_token/" + n for n in rd.vlc_tables() if n.startswith("coeff_token")])
    missing = [k for k in want if not cover[k]]
    assert not missing, missing


def test_golden_streams_stay_within_their_level(decoded):
    """A.3.1 as the independent decoder checks it (Decoder.level_violations: MaxFS and the picture's shape, MaxMBPS
    at the VUI frame rate, 3200 bits per macroblock, the vector ranges, the VUI's own restrictions): every golden
    stream is a conforming stream of the level its SPS names."""
    for case, (_, _, _, _, violations) in decoded:
        if (len(case) > 7 and ("pcm=-1" in case[7] or "roi=" in case[7])) or (ol.PCM_DEFAULT < 0 and case[5] == "C" and case[2] <= 5 and "pcm=0" not in str(case[7:])):
            # I_PCM switched off (these keep the level escapes in the vectors), or a session that does not take it
            # (regions of interest): noise at a low QP passes 3200 bits per macroblock there, and nothing else
            assert all("bits > 3200" in v for v in violations), (case, violations)
        else:
            assert not violations, (case, violations)


def test_level_limits_are_noticed():
    """the checks themselves: a 1080p stream that claims 60 frames/s at level 4.1, and level 1 for 320x240"""
    w, h = 1920, 1080
    for fps, level, want in ((30, 41, None), (60, 41, "MaxMBPS"), (30, 40, None), (30, 32, "MaxFS")):
        d = rd.Decoder()
        d.decode(ol_sps(w, h, level, fps))
        v = d.level_violations()
        assert (not v) if want is None else any(want in x for x in v), (fps, level, v)
    d = rd.Decoder()
    d.decode(ol_sps(320, 240, 10, 0))
    assert any("MaxFS" in x for x in d.level_violations())
    d = rd.Decoder()
    d.decode(ol_sps(3840, 2160, 51, 30))
    assert not d.level_violations()
    d.max_mb_bits, d.max_abs_mv = 3201, [0, 4 * 512 + 1]
    v = d.level_violations()
    assert any("3200" in x for x in v) and any("vertical" in x for x in v)


def ol_sps(w, h, level, fps):
    import ctypes
    cfg = ol.default_config(w, h, level_idc=level, fps_num=fps, fps_den=1 if fps else 0)
    buf = np.zeros(64, np.uint8)
    n = ol.lib().h264o_write_sps(ctypes.byref(cfg), buf.ctypes.data, 64)
    assert n > 0
    return buf[:n].tobytes()


def test_oracle_c_decoder_agrees_with_the_independent_one():
    w, h = 96, 64
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=28, keyint=3, slice_count=2))
    cdec, pdec = ol.OracleDecoder(), rd.Decoder()
    for f in range(5):
        au = oe.encode(*ol.gen_frame("BC"[f & 1], w, h, f))
        assert cdec.decode(au) == 1
        pic, = pdec.decode(au)
        cy, cuv = cdec.picture()
        assert pic.luma_bytes() == np.ascontiguousarray(cy).tobytes()
        assert pic.nv12_chroma_bytes() == np.ascontiguousarray(cuv).tobytes()


def test_a_corrupted_stream_is_noticed():
    w, h = 64, 48
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=30, keyint=25))
    au = bytearray(oe.encode(*ol.gen_frame("B", w, h, 0)))
    good, = rd.Decoder().decode(bytes(au))
    au[len(au) // 2] ^= 0x10
    try:
        bad = rd.Decoder().decode(bytes(au))
    except rd.SyntaxError264:
        return
    assert not bad or bad[0].luma_bytes() != good.luma_bytes()


def test_oracle_reproduces_the_fullsize_golden_access_units():
    """tests/golden/indep_fullsize.json pins the first GOP of BASELINE configs[1..3] (oracle bytes read by the
    independent decoder, tools/indep_fullsize.py). The vectors are only worth something while the oracle still
    emits those bytes: configs[1] whole, the first two access units of configs[2] (the 4K case is left to the tool)."""
    import hashlib
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "indep_fullsize.json")))["cases"]
    for c, n in zip(cases[:2], (25, 2)):
        name, w, h, qp, keyint, frames, slices, fps = c["case"]
        oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=keyint, slice_count=slices, level_idc=c["level_idc"],
                                                fps_num=fps, fps_den=1))
        got = [hashlib.sha1(oe.encode(*ol.gen_frame("B", w, h, f))).hexdigest() for f in range(n)]
        assert got == c["au_sha1"][:n], name
        assert c["max_mb_bits"] <= 3200 and len(c["au_sha1"]) == frames


def test_decoders_agree_on_damaged_streams():
    """tools/fuzz_decoders.py: valid streams with flipped bits, overwritten, inserted and missing bytes, read by the
    oracle's C decoder and by the independent Python decoder. Both refuse a damaged stream, or both accept it and
    rebuild the same pictures (a refusal for a feature one of them does not implement is no verdict)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_decoders as fz
    stats, findings = fz.campaign(seed=5, cases=250)
    assert not findings, findings[:5]
    assert stats["both_refuse"] > 100 and stats["both_accept_same"] > 10, stats
