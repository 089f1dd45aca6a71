"""Whole streams derived BY HAND from ITU-T H.264 (03/2010), bit by bit -- the one pin that none of the three
programs of this repository (oracle/h264o_enc.c, oracle/h264o_dec.c, tests/indep/h264_refdec.py) stands behind.

Every table below is a list of (syntax element, clause, bit string); the strings were written down from the
syntax tables of 7.3, the Exp-Golomb rule of 9.1 and the coeff_token table 9-5, not produced by any encoder.
`_pack` only concatenates them, appends rbsp_trailing_bits (7.3.2.11: a 1, then 0s to the byte boundary) and
puts the start code and NAL header in front (B.1.1, 7.3.1). tests/golden/handmade.json holds the resulting
bytes (`python tests/test_handmade_streams.py --write` rewrites it from these tables alone).

Exp-Golomb, 9.1: codeNum k is written as [M zeros][1][M bits of k + 1 - 2^M], M = floor(log2(k + 1)):
    ue(0) = 1      ue(1) = 010    ue(2) = 011    ue(3) = 00100   ue(4) = 00101   ue(6) = 00111
    ue(7) = 0001000              ue(25) = 000011010             ue(50) = 00000110011
se(v), 9.1.1: codeNum = 2|v| - (v > 0):  se(0) = ue(0) = 1,  se(4) = ue(7),  se(-2) = ue(4),  se(-25) = ue(50).

What the ENCODER decides in these pictures (policy, not standard) is stated where it is used:
  * a flat picture has SAD 0 for every predictor, "I4x4 iff sum SAD4 + 32 lambda < SAD16" is false -> Intra16x16;
    its mode is the available one with least SAD, ties to the lowest mode number: V (0) where the macroblock above
    exists, else H (1) where the one to the left exists, else DC (2); chroma DC (0);
  * a P picture equal to its reference has SAD 0 at vector (0,0), the cheapest vector, no residual -> P_Skip
    everywhere (8.4.1.1: the skip vector is 0 when A or B is missing or has a zero vector with refIdx 0);
  * samples that cannot be coded within 3200 bits (A.3.1) go out as I_PCM.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "handmade.json")


def ue(k):
    """9.1, restated here so that the literal strings below can be checked against the rule"""
    m = (k + 1).bit_length() - 1
    return "0" * m + format(k + 1, "b")


def se(v):
    return ue(2 * abs(v) - (1 if v > 0 else 0))


assert [ue(k) for k in (0, 1, 2, 3, 4, 6, 7, 25, 50)] == ["1", "010", "011", "00100", "00101", "00111", "0001000",
                                                             "000011010", "00000110011"]
assert (se(0), se(4), se(-2), se(-25)) == ("1", "0001000", "00101", "00000110011")


def _bits(fields):
    return "".join(b for _, _, b in fields)


def _pack(nal_header, fields, raw_tail=b""):
    """start code + NAL header byte + the fields + (raw bytes that follow a byte-aligned position) + trailing bits.
    No emulation prevention: the streams below hold no 00 00 0x inside a NAL (asserted)."""
    bits = _bits(fields)
    if raw_tail:
        assert len(bits) % 8 == 0
        body = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)) + raw_tail + b"\x80"
    else:
        bits += "1"
        bits += "0" * (-len(bits) % 8)
        body = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
    payload = bytes([nal_header]) + body
    for i in range(len(payload) - 2):
        assert not (payload[i] == 0 and payload[i + 1] == 0 and payload[i + 2] <= 3), "needs emulation prevention"
    return b"\x00\x00\x00\x01" + payload


# ---- 7.3.2.1.1 seq_parameter_set_data; NAL header 0x67 = forbidden 0, nal_ref_idc 3, nal_unit_type 7 --------------
def sps_fields(mb_w, mb_h):
    return [
        ("profile_idc = 66 (baseline)", "7.3.2.1.1 u(8)", "01000010"),
        ("constraint_set0..5 + reserved_zero_2bits", "u(8)", "00000000"),
        ("level_idc = 41", "u(8)", "00101001"),
        ("seq_parameter_set_id = 0", "ue", "1"),
        ("log2_max_frame_num_minus4 = 0", "ue", "1"),
        ("pic_order_cnt_type = 2", "ue", "011"),
        ("max_num_ref_frames = 1", "ue", "010"),
        ("gaps_in_frame_num_value_allowed_flag", "u(1)", "0"),
        ("pic_width_in_mbs_minus1", "ue", {1: "1", 3: "011"}[mb_w]),
        ("pic_height_in_map_units_minus1", "ue", {1: "1", 2: "010"}[mb_h]),
        ("frame_mbs_only_flag", "u(1)", "1"),
        ("direct_8x8_inference_flag", "u(1)", "0"),
        ("frame_cropping_flag", "u(1)", "0"),
        ("vui_parameters_present_flag", "u(1)", "0"),
    ]


# ---- 7.3.2.2 pic_parameter_set_rbsp; NAL header 0x68 = nal_ref_idc 3, type 8 --------------------------------------
def pps_fields(qp):
    init = {24: "00101",        # se(-2)  = ue(4)
            30: "0001000",      # se(4)   = ue(7)
            1: "00000110011",   # se(-25) = ue(50)
            10: "00000100001",  # se(-16) = ue(32): 33 = 100001b behind five zeros
            36: "000010100"}[qp]  # se(10)  = ue(19): 20 = 10100b behind four zeros
    assert init == se(qp - 26)
    return [
        ("pic_parameter_set_id = 0", "ue", "1"),
        ("seq_parameter_set_id = 0", "ue", "1"),
        ("entropy_coding_mode_flag = 0 (CAVLC)", "u(1)", "0"),
        ("bottom_field_pic_order_in_frame_present_flag", "u(1)", "0"),
        ("num_slice_groups_minus1 = 0", "ue", "1"),
        ("num_ref_idx_l0_default_active_minus1 = 0", "ue", "1"),
        ("num_ref_idx_l1_default_active_minus1 = 0", "ue", "1"),
        ("weighted_pred_flag", "u(1)", "0"),
        ("weighted_bipred_idc", "u(2)", "00"),
        ("pic_init_qp_minus26 = qp - 26", "se", init),
        ("pic_init_qs_minus26 = qp - 26", "se", init),
        ("chroma_qp_index_offset = 4", "se", "0001000"),
        ("deblocking_filter_control_present_flag", "u(1)", "1"),
        ("constrained_intra_pred_flag", "u(1)", "0"),
        ("redundant_pic_cnt_present_flag", "u(1)", "0"),
    ]


# ---- 7.3.3 slice_header -------------------------------------------------------------------------------------------
IDR_SLICE_HEADER = [   # NAL header 0x65 = nal_ref_idc 3, type 5
    ("first_mb_in_slice = 0", "ue", "1"),
    ("slice_type = 2 (I)", "ue", "011"),
    ("pic_parameter_set_id = 0", "ue", "1"),
    ("frame_num = 0", "u(4): log2_max_frame_num = 4", "0000"),
    ("idr_pic_id = 0", "ue", "1"),
    # pic_order_cnt_type 2: no POC syntax. dec_ref_pic_marking() of an IDR picture, 7.3.3.3:
    ("no_output_of_prior_pics_flag", "u(1)", "0"),
    ("long_term_reference_flag", "u(1)", "0"),
    ("slice_qp_delta = 0", "se", "1"),
    ("disable_deblocking_filter_idc = 0", "ue", "1"),
    ("slice_alpha_c0_offset_div2 = 0", "se", "1"),
    ("slice_beta_offset_div2 = 0", "se", "1"),
]
P_SLICE_HEADER = [     # NAL header 0x41 = nal_ref_idc 2, type 1; the picture after the IDR picture
    ("first_mb_in_slice = 0", "ue", "1"),
    ("slice_type = 0 (P)", "ue", "1"),
    ("pic_parameter_set_id = 0", "ue", "1"),
    ("frame_num = 1", "u(4)", "0001"),
    ("num_ref_idx_active_override_flag", "u(1)", "0"),
    ("ref_pic_list_modification_flag_l0", "u(1)", "0"),
    ("adaptive_ref_pic_marking_mode_flag", "u(1), nal_ref_idc != 0", "0"),
    ("slice_qp_delta = 0", "se", "1"),
    ("disable_deblocking_filter_idc = 0", "ue", "1"),
    ("slice_alpha_c0_offset_div2 = 0", "se", "1"),
    ("slice_beta_offset_div2 = 0", "se", "1"),
]
assert len(_bits(IDR_SLICE_HEADER)) == 16 and len(_bits(P_SLICE_HEADER)) == 14


# ---- 7.3.5 macroblock_layer of a flat Intra16x16 macroblock -------------------------------------------------------
def flat_i16_mb(pred_mode):
    """Table 7-11: mb_type of I_16x16_<pred>_<chroma cbp>_<luma cbp> = 1 + pred + 4 * chroma + 12 * (luma != 0); here
    both patterns are 0 (the residual of a flat block predicted from flat neighbours, or from 128, is 0)."""
    return [
        ("mb_type = 1 + pred_mode", "ue, Table 7-11", {0: "010", 1: "011", 2: "00100"}[pred_mode]),
        # mb_pred(): Intra16x16 has no per-block modes; only
        ("intra_chroma_pred_mode = 0 (DC)", "ue", "1"),
        # no coded_block_pattern for Intra16x16 (it is in mb_type); mb_qp_delta is always there for Intra16x16
        ("mb_qp_delta = 0", "se", "1"),
        # residual(): Intra16x16DCLevel is always coded. nC (9.2.1): the neighbouring blocks A / B are missing or hold
        # no AC level -> nC = 0; TotalCoeff 0, TrailingOnes 0 in the column 0 <= nC < 2 of Table 9-5: "1"
        ("coeff_token(TotalCoeff 0, TrailingOnes 0), nC = 0", "ce, Table 9-5", "1"),
        # CodedBlockPatternLuma = 0: no Intra16x16ACLevel; CodedBlockPatternChroma = 0: no chroma
    ]


def dc_only_i16_mb(qp):
    """One macroblock without neighbours, luma flat 128 + d, chroma 128. Decision (policy, from source samples): the
    only Intra16x16 predictor is DC = 128, SAD16 = 256 d; Intra4x4 would pay 16 d for its first block (DC = 128), 0 for
    the fifteen that lean on source neighbours, plus 32 lambda: Intra4x4 iff 16 d + 32 lambda < 256 d. With d = 1 at
    QP 30 (lambda 8: 272 < 256 is false) and d = 2 at QP 36 (lambda 16: 544 < 512 is false) it stays Intra16x16.
    Intra16x16 DC predicts 128 (8.3.3.3), the residual is d everywhere. Forward core transform of a flat 4x4 block:
    only W00 = 16 d. The 4x4 Hadamard of the sixteen DCs (all 16 d): only its (0,0) = 256 d, halved with rounding:
    128 d. Quantiser (policy, DESIGN.md 3: JM dead zone, intra f = 2^qbits / 3; the DC uses 2 f and qbits + 1;
    qbits = 15 + QP / 6; MF(QP % 6 = 0, position (0,0)) = 13107):
        QP 30, d 1: (128 * 13107 + 2 * (2^20 / 3)) >> 21 = (1677696 + 699050) >> 21 = 1
        QP 36, d 2: (256 * 13107 + 2 * (2^21 / 3)) >> 22 = (3355392 + 1398100) >> 22 = 1
    residual_block_cavlc(Intra16x16DCLevel), 9.2: TotalCoeff 1, TrailingOnes 1, nC 0 -> coeff_token 01 (Table 9-5);
    trailing_ones_sign_flag 0 (+1); total_zeros = 0 with one coefficient (Table 9-7, tzVlcIndex 1): 1.
    The decoder (8.5.10, 8.5.12): inverse Hadamard of the lone 1 = 1 at all sixteen positions; LevelScale(0,0,0) =
    16 * 10 = 160. QP 30: dcY = (1 * 160 + 1) >> 1 = 80, every sample 128 + ((80 + 32) >> 6) = 129: exact.
    QP 36: dcY = (1 * 160) << 0 = 160, 128 + ((160 + 32) >> 6) = 131, one above the source. The second picture (the same
    source): QP 30 its reference equals it; QP 36 the residual is -1 everywhere, |W00| = 16 per block, inter dead zone
    (16 * 13107 + 2^21 / 6) >> 21 = 0: P_Skip either way, the reconstruction stays 129 resp. 131."""
    return [
        ("mb_type = 3: I_16x16_2_0_0 (DC; AC and chroma patterns 0)", "ue, Table 7-11", "00100"),
        ("intra_chroma_pred_mode = 0 (DC)", "ue", "1"),
        ("mb_qp_delta = 0", "se", "1"),
        ("coeff_token(TotalCoeff 1, TrailingOnes 1), nC = 0", "ce, Table 9-5", "01"),
        ("trailing_ones_sign_flag = 0", "u(1)", "0"),
        ("total_zeros = 0, tzVlcIndex 1", "ce, Table 9-7", "1"),
    ]


def flat_picture_slice_data(mb_w, mb_h):
    f = []
    for y in range(mb_h):
        for x in range(mb_w):
            f += flat_i16_mb(0 if y > 0 else 1 if x > 0 else 2)  # V where there is a row above, else H, else DC
    return f


def skip_picture_slice_data(n_mbs):
    # 7.3.4: mb_skip_run = every macroblock of the picture, then no more data
    return [("mb_skip_run = %d" % n_mbs, "ue", {1: "010", 6: "00111"}[n_mbs])]


def pcm_samples():
    """16x16 luma, 8x8 Cb, 8x8 Cr: a pattern no transform coder gets under 3200 bits at a fine quantiser, and without
    a byte below 4, so that no 00 00 0x can occur inside the NAL."""
    i = np.arange(256)
    y = (16 + (i * 97 + (i // 16) * 59 + (i * i) % 251) % 224).astype(np.uint8)
    j = np.arange(64)
    cb = (16 + (j * 83 + (j * j) % 127) % 224).astype(np.uint8)
    cr = (16 + (j * 61 + (j * j * 3) % 113) % 224).astype(np.uint8)
    return y, cb, cr


PCM_MB = [
    ("mb_type = 25 (I_PCM)", "ue, Table 7-11", "000011010"),
    # 16 bits of slice header + 9 = bit 25: seven pcm_alignment_zero_bit reach the byte boundary (7.3.5)
    ("pcm_alignment_zero_bit x 7", "f(1)", "0000000"),
    # then pcm_sample_luma[256] in raster order, pcm_sample_chroma[128]: all of Cb, then all of Cr, u(8) each
]


def streams():
    """name -> (config, frames as (y, cb, cr) planes, access units as bytes)"""
    out = {}
    for qp in (24, 30):
        for (mw, mh) in ((1, 1), (3, 2)):
            w, h = 16 * mw, 16 * mh
            flat = (np.full(w * h, 128, np.uint8), np.full(w * h // 4, 128, np.uint8), np.full(w * h // 4, 128, np.uint8))
            au0 = (_pack(0x67, sps_fields(mw, mh)) + _pack(0x68, pps_fields(qp)) +
                   _pack(0x65, IDR_SLICE_HEADER + flat_picture_slice_data(mw, mh)))
            au1 = _pack(0x41, P_SLICE_HEADER + skip_picture_slice_data(mw * mh))
            out["flat_grey_%dx%d_qp%d" % (w, h, qp)] = dict(width=w, height=h, qp=qp, frames=[flat, flat], aus=[au0, au1])
    for qp, src, rec in ((30, 129, 129), (36, 130, 131)):
        f140 = (np.full(256, src, np.uint8), np.full(64, 128, np.uint8), np.full(64, 128, np.uint8))
        au0 = _pack(0x67, sps_fields(1, 1)) + _pack(0x68, pps_fields(qp)) + _pack(0x65, IDR_SLICE_HEADER + dc_only_i16_mb(qp))
        au1 = _pack(0x41, P_SLICE_HEADER + skip_picture_slice_data(1))
        out["dc_only_16x16_qp%d" % qp] = dict(width=16, height=16, qp=qp, frames=[f140, f140], aus=[au0, au1],
                                                 recon=[(np.full(256, rec, np.uint8), f140[1], f140[2])] * 2)
    for qp in (1, 10):
        y, cb, cr = pcm_samples()
        au0 = (_pack(0x67, sps_fields(1, 1)) + _pack(0x68, pps_fields(qp)) +
               _pack(0x65, IDR_SLICE_HEADER + PCM_MB, raw_tail=y.tobytes() + cb.tobytes() + cr.tobytes()))
        # the reconstruction of an I_PCM macroblock IS its samples (8.3.5), so a second, identical picture is all skip
        au1 = _pack(0x41, P_SLICE_HEADER + skip_picture_slice_data(1))
        out["i_pcm_16x16_qp%d" % qp] = dict(width=16, height=16, qp=qp, frames=[(y, cb, cr)] * 2, aus=[au0, au1])
    return out


def nv12(frame):
    y, cb, cr = frame
    uv = np.empty(cb.size * 2, np.uint8)
    uv[0::2], uv[1::2] = cb, cr
    return y, uv


def _golden():
    return json.load(open(GOLDEN))


def test_the_json_holds_exactly_what_the_tables_say():
    g = _golden()["streams"]
    s = streams()
    assert set(g) == set(s)
    for name, v in s.items():
        assert [bytes.fromhex(a) for a in g[name]["access_units"]] == v["aus"], name


def test_a_few_bytes_checked_by_eye():
    """what the first stream looks like in hex, as derived on paper in DESIGN.md 2 (round 4)"""
    s = streams()["flat_grey_16x16_qp24"]
    assert s["aus"][0].hex(" ") == ("00 00 00 01 67 42 00 29 da 71 "      # SPS: 1 1 011 010 0 1 1 1 0 0 0 | stop bit 1
                                    "00 00 00 01 68 ce 0a 51 12 "         # PPS (= tests/golden/headers.json cavlc_qp24)
                                    "00 00 00 01 65 b8 4f 27 80")         # slice header b8 4f; 00100 1 1 1 = 27; stop bit
    assert s["aus"][1].hex(" ") == "00 00 00 01 41 e2 3d 40"              # 14 header bits, 010, stop bit


@pytest.mark.parametrize("name", sorted(streams()))
def test_the_oracle_emits_the_handmade_streams(name):
    from tests import oracle_lib as ol
    v = streams()[name]
    enc = ol.OracleEncoder(ol.default_config(v["width"], v["height"], qp=v["qp"], keyint=25))
    dec = ol.OracleDecoder()
    for f, r, au in zip(v["frames"], v.get("recon", v["frames"]), v["aus"]):
        y, uv = nv12(f)
        assert enc.encode(y, uv) == au, name
        assert dec.decode(au) == 1
        dy, duv = dec.picture()
        ry, ruv = nv12(r)  # flat / I_PCM: lossless; the DC-only pictures: what the docstring of dc_only_i16_mb derives
        assert np.array_equal(dy.reshape(-1), ry) and np.array_equal(duv.reshape(-1), ruv)


@pytest.mark.parametrize("name", sorted(streams()))
def test_the_independent_decoder_reads_the_handmade_streams(name):
    from tests.indep import h264_refdec as rd
    v = streams()[name]
    d = rd.Decoder()
    for au, f in zip(v["aus"], v.get("recon", v["frames"])):
        pic, = d.decode(au)
        y, uv = nv12(f)
        assert pic.luma_bytes() == y.tobytes() and pic.nv12_chroma_bytes() == uv.tobytes()
    assert not d.level_violations()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(streams()))
def test_the_hip_path_emits_the_handmade_streams(name):
    import cedarx_h264_encoder_amd as ce
    v = streams()[name]
    with ce.CedarEncoder(width=v["width"], height=v["height"], qp=v["qp"], keyframe_interval=25, device=0) as enc:
        for f, au in zip(v["frames"], v["aus"]):
            y, uv = nv12(f)
            assert enc.encode_frame(y, uv) == au, name


if __name__ == "__main__" and "--write" in sys.argv:
    doc = {"_source": "derived by hand from ITU-T H.264 in tests/test_handmade_streams.py (no encoder involved); "
                      "frames: flat grey 128 / the formula of pcm_samples()",
           "streams": {k: {"width": v["width"], "height": v["height"], "qp": v["qp"],
                           "access_units": [a.hex() for a in v["aus"]]} for k, v in streams().items()}}
    json.dump(doc, open(GOLDEN, "w"), indent=1)
    print("wrote", GOLDEN)
