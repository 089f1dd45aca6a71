"""Structural checks on the spec tables: no external H.264
decoder or copy of the standard exists in the environment, so every VLC table
must at least be a prefix-free code (complete where the spec's is), the quant
and dequant tables must be reciprocal, and the deblock tables monotone."""
import ctypes
import re
from fractions import Fraction

import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _parse_tables(path):
    """Pull `static const uintN_t name[..]... = { ... };` arrays out of a C header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"static const u?int\d+_t (\w+)((?:\[\d*\])+)\s*=\s*\{(.*?)\};", src, re.S):
        name, dims, body = m.group(1), [int(d) for d in re.findall(r"\[(\d+)\]", m.group(2))], m.group(3)
        # split top-level rows when 2-D
        if len(dims) == 2:
            rows = re.findall(r"\{([^{}]*)\}", body)
            arr = np.zeros(dims, np.int64)
            for i, r in enumerate(rows):
                vals = [int(x) for x in re.findall(r"-?\d+", r)]
                arr[i, :len(vals)] = vals
            out[name] = arr
        else:
            vals = [int(x) for x in re.findall(r"-?\d+", body)]
            arr = np.zeros(dims[0], np.int64)
            arr[:len(vals)] = vals
            out[name] = arr
    return out


T = _parse_tables(os.path.join(ROOT, "oracle", "h264o_tables.h"))


def _codes(lens, bits):
    return [(int(l), int(b)) for l, b in zip(lens, bits) if l]


def _prefix_free(codes):
    strs = [format(b, "0%db" % l) for l, b in codes]
    assert all(b < (1 << l) for l, b in codes)
    assert len(set(strs)) == len(strs)
    for i, a in enumerate(strs):
        for j, b in enumerate(strs):
            if i != j:
                assert not b.startswith(a), (a, b)


def _kraft(codes):
    return sum(Fraction(1, 1 << l) for l, _ in codes)


@pytest.mark.parametrize("tab", range(4))
def test_coeff_token_prefix_free(tab):
    codes = _codes(T["h264o_coeff_token_len"][tab], T["h264o_coeff_token_bits"][tab])
    assert len(codes) == 62  # 1 + 2 + 3 + 14*4
    _prefix_free(codes)
    assert _kraft(codes) <= 1


def test_coeff_token_validity_mask():
    for tab in range(4):
        lens = T["h264o_coeff_token_len"][tab]
        for tc in range(17):
            for t1 in range(4):
                valid = t1 <= tc and t1 <= 3
                assert (lens[4 * tc + t1] != 0) == valid


def test_chroma_dc_coeff_token():
    codes = _codes(T["h264o_chroma_dc_coeff_token_len"], T["h264o_chroma_dc_coeff_token_bits"])
    assert len(codes) == 14
    _prefix_free(codes)


@pytest.mark.parametrize("tc", range(1, 16))
def test_total_zeros_complete_prefix_code(tc):
    lens, bits = T["h264o_total_zeros_len"][tc - 1], T["h264o_total_zeros_bits"][tc - 1]
    codes = _codes(lens, bits)
    assert len(codes) == 16 - tc + 1
    assert all(lens[i] for i in range(16 - tc + 1))
    _prefix_free(codes)
    assert _kraft(codes) == 1 or tc == 1  # table for tc=1 is the one incomplete table (all-zero prefix unused)
    if tc == 1:
        assert _kraft(codes) == 1 - Fraction(1, 512)


@pytest.mark.parametrize("tc", range(1, 4))
def test_chroma_dc_total_zeros(tc):
    codes = _codes(T["h264o_chroma_dc_total_zeros_len"][tc - 1], T["h264o_chroma_dc_total_zeros_bits"][tc - 1])
    assert len(codes) == 4 - tc + 1
    _prefix_free(codes)
    assert _kraft(codes) == 1


@pytest.mark.parametrize("zl", range(1, 8))
def test_run_before(zl):
    codes = _codes(T["h264o_run_before_len"][zl - 1], T["h264o_run_before_bits"][zl - 1])
    assert len(codes) == (zl + 1 if zl < 7 else 15)
    _prefix_free(codes)
    if zl < 7:
        assert _kraft(codes) == 1


def test_cbp_maps_are_permutations():
    for n in ("h264o_cbp_intra_from_code", "h264o_cbp_inter_from_code"):
        assert sorted(T[n].tolist()) == list(range(48))
    # most probable patterns get the shortest codes
    assert T["h264o_cbp_intra_from_code"][0] == 47 and T["h264o_cbp_inter_from_code"][0] == 0


def test_quant_dequant_reciprocal():
    """MF * V ~= 2^21 * {1/16, 1/25, 1/20} per class."""
    mf, v = T["h264o_quant_mf"], T["h264o_dequant_v"]
    norm = [Fraction(1, 16), Fraction(1, 25), Fraction(1, 20)]
    for m in range(6):
        for c in range(3):
            target = float((1 << 21) * norm[c])
            assert abs(mf[m][c] * v[m][c] - target) / target < 2e-3
    assert (mf[0] == [13107, 5243, 8066]).all() and (v[0] == [10, 16, 13]).all()


def test_pos_class_and_zigzag():
    z = T["h264o_zigzag4x4"].tolist()
    assert sorted(z) == list(range(16))
    # zig-zag: successive positions are 4-neighbours or diagonal neighbours
    for a, b in zip(z, z[1:]):
        assert max(abs(a // 4 - b // 4), abs(a % 4 - b % 4)) == 1
    pc = T["h264o_pos_class"]
    for i in range(16):
        r, c = i // 4, i % 4
        want = 0 if (r % 2 == 0 and c % 2 == 0) else 1 if (r % 2 and c % 2) else 2
        assert pc[i] == want


def test_block_order_tables():
    bx, by = T["h264o_blk_x"], T["h264o_blk_y"]
    assert sorted((int(x), int(y)) for x, y in zip(bx, by)) == [(x, y) for x in range(4) for y in range(4)]
    for i in range(16):  # 8x8 quadrant then 4x4 inside it
        assert bx[i] == 2 * ((i >> 2) & 1) + (i & 1) and by[i] == 2 * (i >> 3) + ((i >> 1) & 1)


def test_deblock_tables_monotone():
    a, b, tc = T["h264o_alpha"], T["h264o_beta"], T["h264o_tc0"]
    assert (np.diff(a) >= 0).all() and (np.diff(b) >= 0).all()
    assert (np.diff(tc, axis=0) >= 0).all() and (np.diff(tc, axis=1) >= 0).all()
    assert a[15] == 0 and a[16] == 4 and a[51] == 255 and b[16] == 2 and b[51] == 18
    assert tc[51].tolist() == [13, 17, 25] and not tc[:17].any()


def test_chroma_qp_table():
    q = T["h264o_chroma_qp"]
    assert (q[:30] == np.arange(30)).all() and q[51] == 39 and (np.diff(q) >= 0).all() and (np.diff(q) <= 1).all()


def test_lambda_table():
    lam = T["h264o_lambda"]
    for qp in range(52):
        assert lam[qp] == max(1, int(2 ** ((qp - 12) / 6) + 0.5))
