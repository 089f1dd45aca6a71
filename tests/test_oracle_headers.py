"""Oracle vs the reference-derived known-answer vectors.

These are the only results the reference pins for this path: the bytes its
header writer (kernel/cedar.c:187-223, :868-1030) produces.
"""
import ctypes
import json
import os

import numpy as np
import pytest

from tests import oracle_lib as ol

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "headers.json")))


def _hex(s):
    return bytes(int(x, 16) for x in s.split())


def _sps(c):
    buf = np.zeros(64, np.uint8)
    n = ol.lib().h264o_write_sps(ctypes.byref(c), buf.ctypes.data, 64)
    return buf[:n].tobytes()


def _pps(c):
    buf = np.zeros(64, np.uint8)
    n = ol.lib().h264o_write_pps(ctypes.byref(c), buf.ctypes.data, 64)
    return buf[:n].tobytes()


@pytest.mark.parametrize("v", G["sps"], ids=lambda v: v["name"])
def test_sps_known_answers(v):
    c = ol.default_config(v["width"], v["height"], profile_idc=v["profile"], level_idc=v["level"])
    assert _sps(c) == _hex(v["hex"])


@pytest.mark.parametrize("v", G["sps_reference_quirk"], ids=lambda v: v["name"])
def test_sps_reference_trailing_quirk_is_the_only_difference(v):
    c = ol.default_config(v["width"], v["height"], profile_idc=v["profile"], level_idc=v["level"])
    got = _sps(c)
    assert got == _hex(v["standard_hex"])
    assert _hex(v["hex"]) == got + b"\x00"  # cedar.c:883-890 stray byte, documented divergence


@pytest.mark.parametrize("v", G["pps"], ids=lambda v: v["name"])
def test_pps_known_answers(v):
    c = ol.default_config(320, 240, qp=v["qp"])
    assert _pps(c) == _hex(v["hex"])


def test_pps_differs_from_reference_literal_only_in_entropy_flag():
    v = G["pps_reference_literal"][0]
    got = bytearray(_pps(ol.default_config(320, 240, qp=v["qp"])))
    ref = _hex(v["hex"])
    assert len(got) == len(ref)
    diff = [i for i in range(len(ref)) if got[i] != ref[i]]
    assert diff == [5] and got[5] ^ ref[5] == 0x20  # entropy_coding_mode_flag bit


@pytest.mark.parametrize("v", G["slice_header"], ids=lambda v: v["name"])
def test_slice_header_known_answers(v):
    c = ol.default_config(320, 240)
    buf = np.zeros(64, np.uint8)
    nbits = ctypes.c_int()
    n = ol.lib().h264o_write_slice_header(ctypes.byref(c), v["idr"], v["frame_num"], v["idr_pic_id"], 0,
                                          buf.ctypes.data, 64, ctypes.byref(nbits))
    got = buf[:n].tobytes()
    want = _hex(v["hex"])
    assert nbits.value == v["nbits"]
    assert got[:len(want)] == want
    if "tail_bits" in v:
        tail = format(got[len(want)], "08b")[:len(v["tail_bits"])]
        assert tail == v["tail_bits"]


@pytest.mark.parametrize("v", G["cavlc_block"], ids=lambda v: v["name"])
def test_cavlc_block_known_answer(v):
    lv = np.array(v["levels_zigzag"], np.int16)
    out = np.zeros(64, np.uint8)
    n = ol.lib().h264o_cavlc_block(lv.ctypes.data, v["max_coeff"], v["nC"], out.ctypes.data, 64)
    bits = "".join(format(b, "08b") for b in out)[:n]
    assert bits == v["bits"]


def test_transform_known_answers():
    """constant block -> only W[0][0] = 16c; DC-only d -> (d+32)>>6 everywhere."""
    for cst in (1, -3, 100):
        x = np.full(16, cst, np.int16)
        w = np.zeros(16, np.int16)
        ol.lib().h264o_fdct4x4(x.ctypes.data, w.ctypes.data)
        assert w[0] == 16 * cst and not w[1:].any()
    for d in (64, 640, -640, 100):
        coef = np.zeros(16, np.int16)
        coef[0] = d
        dst = np.full(16, 128, np.uint8)
        ol.lib().h264o_idct4x4_add(coef.ctypes.data, dst.ctypes.data, 4)
        assert (dst == np.clip(128 + ((d + 32) >> 6), 0, 255)).all()


def test_config_validation_follows_reference_rules():
    """kernel/cedar.c:749-789: odd sizes, QP outside 1..47, keyint outside 1..31 -> EINVAL."""
    import errno
    for kw in ({"width": 321}, {"height": 241}, {"qp": 0}, {"qp": 48}, {"keyint": 0}, {"keyint": 32}):
        base = dict(width=320, height=240)
        base.update({k: v for k, v in kw.items() if k in ("width", "height")})
        c = ol.default_config(base["width"], base["height"], **{k: v for k, v in kw.items() if k not in base})
        with pytest.raises(OSError) as e:
            ol.OracleEncoder(c)
        assert e.value.errno == errno.EINVAL


def test_gop_cadence_and_frame_num():
    """cedar.c:1047-1050, :1193-1196: I P..P repeating; SPS/PPS only before frame 0."""
    c = ol.default_config(32, 32, keyint=3, qp=30)
    e = ol.OracleEncoder(c)
    y, uv = ol.gen_frame("A", 32, 32, 0)
    types = []
    for f in range(7):
        au = e.encode(y, uv)
        nal_types = [au[i + 4] & 0x1f for i in range(len(au) - 4) if au[i:i + 4] == b"\0\0\0\1"]
        types.append(nal_types)
    assert types[0] == [7, 8, 5]
    assert types[1:] == [[1], [1], [5], [1], [1], [5]]


def test_repeat_headers_puts_parameter_sets_before_every_idr():
    c = ol.default_config(32, 32, keyint=2, qp=30, repeat_headers=1)
    e = ol.OracleEncoder(c)
    y, uv = ol.gen_frame("A", 32, 32, 0)
    kinds = []
    for f in range(4):
        au = e.encode(y, uv)
        kinds.append([au[i + 4] & 0x1f for i in range(len(au) - 4) if au[i:i + 4] == b"\0\0\0\1"])
    assert kinds == [[7, 8, 5], [1], [7, 8, 5], [1]]


# ---- header extensions next to the path: access unit delimiter, VUI timing ----
class _Bits:
    """Independent restatement of the SPS syntax (H.264 7.3.2.1.1 + E.1.1) for the known answers."""

    def __init__(self):
        self.b = []

    def u(self, v, n):
        self.b += [(v >> i) & 1 for i in range(n - 1, -1, -1)]

    def ue(self, v):
        v += 1
        self.u(v, 2 * v.bit_length() - 1)

    def nal(self, hdr):
        bits = self.b + [1]
        bits += [0] * (-len(bits) % 8)
        rbsp = bytes(int("".join(map(str, bits[i:i + 8])), 2) for i in range(0, len(bits), 8))
        out, zeros = bytearray(b"\0\0\0\1" + bytes([hdr])), 0
        for x in rbsp:
            if zeros == 2 and x <= 3:
                out.append(3)
                zeros = 0
            out.append(x)
            zeros = zeros + 1 if x == 0 else 0
        return bytes(out)


def _sps_with_vui(w, h, profile, level, num, den):
    cw, ch = (w + 15) & ~15, (h + 15) & ~15
    b = _Bits()
    b.u(profile, 8), b.u(0, 8), b.u(level, 8)
    b.ue(0), b.ue(0), b.ue(2), b.ue(1), b.u(0, 1)
    b.ue(cw // 16 - 1), b.ue(ch // 16 - 1), b.u(1, 1), b.u(0, 1)
    if (cw, ch) != (w, h):
        b.u(1, 1), b.ue(0), b.ue((cw - w) // 2), b.ue(0), b.ue((ch - h) // 2)
    else:
        b.u(0, 1)
    b.u(1, 1)                                   # vui_parameters_present_flag
    b.u(0, 4), b.u(1, 1), b.u(den, 32), b.u(2 * num, 32), b.u(1, 1)
    b.u(0, 3), b.u(1, 1), b.u(1, 1), b.ue(0), b.ue(0), b.ue(7), b.ue(7), b.ue(0), b.ue(1)
    return b.nal(0x67)


@pytest.mark.parametrize("w,h,num,den", [(320, 240, 30, 1), (854, 480, 30000, 1001), (1920, 1080, 25, 1),
                                         (3840, 2160, 60, 1), (64, 48, 1, 1)])
def test_sps_vui_timing_matches_independent_writer(w, h, num, den):
    c = ol.default_config(w, h, fps_num=num, fps_den=den)
    got = _sps(c)
    assert got == _sps_with_vui(w, h, c.profile_idc, c.level_idc, num, den)
    # num_units_in_tick = 1 puts 00 00 00 01 into the RBSP: emulation prevention must break it up
    assert b"\0\0\0\1" not in got[4:] and b"\0\0\1" not in got[4:]
    # and without a frame rate the SPS is the reference's (vui_parameters_present_flag = 0)
    assert _sps(ol.default_config(w, h)) != got


def test_sps_vui_golden_bytes():
    v = G["sps_vui"][0]
    c = ol.default_config(v["width"], v["height"], fps_num=v["fps_num"], fps_den=v["fps_den"])
    assert _sps(c) == _hex(v["hex"])


def test_access_unit_delimiter_bytes():
    buf = np.zeros(16, np.uint8)
    for idr, want in ((1, "00 00 00 01 09 10"), (0, "00 00 00 01 09 30")):
        n = ol.lib().h264o_write_aud(idr, buf.ctypes.data, 16)
        assert buf[:n].tobytes() == _hex(want)  # nal_ref_idc 0, type 9; primary_pic_type 0 / 1 + stop bit


def test_frame_rate_validation():
    for bad in (dict(fps_num=30), dict(fps_den=1), dict(fps_num=-1, fps_den=1), dict(fps_num=1 << 30, fps_den=1)):
        with pytest.raises(Exception):
            ol.OracleEncoder(ol.default_config(64, 48, **bad))


def test_stream_with_aud_and_vui_decodes_and_reports_timing():
    w, h = 96, 64
    enc = ol.OracleEncoder(ol.default_config(w, h, qp=28, keyint=3, aud=1, fps_num=30000, fps_den=1001,
                                             repeat_headers=1, slice_count=2))
    plain = ol.OracleEncoder(ol.default_config(w, h, qp=28, keyint=3, repeat_headers=1, slice_count=2))
    dec = ol.OracleDecoder()
    for f in range(7):
        y, uv = ol.gen_frame("B", w, h, f)
        au, ref = enc.encode(y, uv), plain.encode(y, uv)
        kinds = [au[i + 4] & 0x1f for i in range(len(au) - 4) if au[i:i + 4] == b"\0\0\0\1"]
        assert kinds == ([9, 7, 8, 5, 5] if f % 3 == 0 else [9, 1, 1])
        # the delimiter and the VUI change nothing below the parameter sets
        assert au[au.index(b"\0\0\0\1\x65" if f % 3 == 0 else b"\0\0\0\1\x41"):] == \
            ref[ref.index(b"\0\0\0\1\x65" if f % 3 == 0 else b"\0\0\0\1\x41"):]
        assert dec.decode(au) == 1
        ry, ruv = enc.recon()
        dy, duv = dec.picture()
        assert np.array_equal(dy, ry) and np.array_equal(duv, ruv)
    assert dec.frame_rate() == (60000, 2002) and dec.reorder_frames() == 0 and dec.aud_count() == 7
