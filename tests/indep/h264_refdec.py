"""An independent H.264 baseline / CAVLC decoder in plain Python, written from the text of
ITU-T Rec. H.264 (clauses 7.3, 7.4, 8.3, 8.4, 8.5, 8.7, 9.1, 9.2) for ONE purpose: to read the
streams this repository's encoders emit with code and tables that share NOTHING with oracle/
(the C oracle and its decoder) or with cedarx_h264_encoder_amd/csrc/ (the product).

TEST INFRASTRUCTURE. It exists because the reference (libv/cedarx_h264_encoder) holds no vectors
and no software for the macroblock stages (they are silicon, kernel/cedar.c:1176-1180), and
because the build and GPU boxes carry no third-party decoder. Every table
below was typed in for this file, in a different layout from oracle/h264o_tables.h (code length +
code value pairs per symbol, the way the standard prints them), and tests/test_independent_decoder.py
first checks them structurally (each VLC table prefix-free and Kraft-complete).

Scope = what the encoders emit plus what is adjacent: I and P slices, frame pictures, one reference
frame, Intra4x4 / Intra16x16 / I_PCM, P_L0_16x16 / P_Skip (other inter partitions are refused),
quarter-sample luma and eighth-sample chroma interpolation, the in-loop deblocking filter with
disable_deblocking_filter_idc 0 / 1 / 2, several slices per picture, frame cropping is reported
but pictures are kept at coded size. Slow (pure Python, ~1000 macroblocks/s).

The decoder also counts which parts of the standard a stream exercised (`Decoder.cover`) so a
test can insist that its streams reached every intra predictor, every total_zeros / run_before
row, the level escape, every boundary strength, and so on.
"""
from collections import Counter

# ----------------------------------------------------------------------------------------
# Tables (clause 9.2 CAVLC, Table 9-4 me(v), 8.5 scaling, 8.7 filter thresholds)
# ----------------------------------------------------------------------------------------

# Table 9-5, coeff_token. One row per TotalCoeff 0..16, four entries per row = TrailingOnes 0..3,
# entry = (length, code value) or None where the combination does not exist.
_N = None
COEFF_TOKEN_0 = [  # 0 <= nC < 2
    [(1, 1), _N, _N, _N],
    [(6, 5), (2, 1), _N, _N],
    [(8, 7), (6, 4), (3, 1), _N],
    [(9, 7), (8, 6), (7, 5), (5, 3)],
    [(10, 7), (9, 6), (8, 5), (6, 3)],
    [(11, 7), (10, 6), (9, 5), (7, 4)],
    [(13, 15), (11, 6), (10, 5), (8, 4)],
    [(13, 11), (13, 14), (11, 5), (9, 4)],
    [(13, 8), (13, 10), (13, 13), (10, 4)],
    [(14, 15), (14, 14), (13, 9), (11, 4)],
    [(14, 11), (14, 10), (14, 13), (13, 12)],
    [(15, 15), (15, 14), (14, 9), (14, 12)],
    [(15, 11), (15, 10), (15, 13), (14, 8)],
    [(16, 15), (15, 1), (15, 9), (15, 12)],
    [(16, 11), (16, 14), (16, 13), (15, 8)],
    [(16, 7), (16, 10), (16, 9), (16, 12)],
    [(16, 4), (16, 6), (16, 5), (16, 8)],
]
COEFF_TOKEN_2 = [  # 2 <= nC < 4
    [(2, 3), _N, _N, _N],
    [(6, 11), (2, 2), _N, _N],
    [(6, 7), (5, 7), (3, 3), _N],
    [(7, 7), (6, 10), (6, 9), (4, 5)],
    [(8, 7), (6, 6), (6, 5), (4, 4)],
    [(8, 4), (7, 6), (7, 5), (5, 6)],
    [(9, 7), (8, 6), (8, 5), (6, 8)],
    [(11, 15), (9, 6), (9, 5), (6, 4)],
    [(11, 11), (11, 14), (11, 13), (7, 4)],
    [(12, 15), (11, 10), (11, 9), (9, 4)],
    [(12, 11), (12, 14), (12, 13), (11, 12)],
    [(12, 8), (12, 10), (12, 9), (11, 8)],
    [(13, 15), (13, 14), (13, 13), (12, 12)],
    [(13, 11), (13, 10), (13, 9), (13, 12)],
    [(13, 7), (14, 11), (13, 6), (13, 8)],
    [(14, 9), (14, 8), (14, 10), (13, 1)],
    [(14, 7), (14, 6), (14, 5), (14, 4)],
]
COEFF_TOKEN_4 = [  # 4 <= nC < 8
    [(4, 15), _N, _N, _N],
    [(6, 15), (4, 14), _N, _N],
    [(6, 11), (5, 15), (4, 13), _N],
    [(6, 8), (5, 12), (5, 14), (4, 12)],
    [(7, 15), (5, 10), (5, 11), (4, 11)],
    [(7, 11), (5, 8), (5, 9), (4, 10)],
    [(7, 9), (6, 14), (6, 13), (4, 9)],
    [(7, 8), (6, 10), (6, 9), (4, 8)],
    [(8, 15), (7, 14), (7, 13), (5, 13)],
    [(8, 11), (8, 14), (7, 10), (6, 12)],
    [(9, 15), (8, 10), (8, 13), (7, 12)],
    [(9, 11), (9, 14), (8, 9), (8, 12)],
    [(9, 8), (9, 10), (9, 13), (8, 8)],
    [(10, 13), (9, 7), (9, 9), (9, 12)],
    [(10, 9), (10, 12), (10, 11), (10, 10)],
    [(10, 5), (10, 8), (10, 7), (10, 6)],
    [(10, 1), (10, 4), (10, 3), (10, 2)],
]
# 8 <= nC: six bits, xxxxyy with xxxx = TotalCoeff - 1 and yy = TrailingOnes, except 000011 = (0, 0)
COEFF_TOKEN_8 = [[None] * 4 for _ in range(17)]
COEFF_TOKEN_8[0][0] = (6, 3)
for _tc in range(1, 17):
    for _t1 in range(0, min(_tc, 3) + 1):
        COEFF_TOKEN_8[_tc][_t1] = (6, ((_tc - 1) << 2) | _t1)
COEFF_TOKEN_CDC = [  # nC == -1, chroma DC of 4:2:0 (TotalCoeff 0..4)
    [(2, 1), _N, _N, _N],
    [(6, 7), (1, 1), _N, _N],
    [(6, 4), (6, 6), (3, 1), _N],
    [(6, 3), (7, 3), (7, 2), (6, 5)],
    [(6, 2), (8, 3), (8, 2), (7, 0)],
]

# Tables 9-7 / 9-8: total_zeros for 4x4 blocks; row = tzVlcIndex (TotalCoeff) 1..15, entry k = total_zeros k
TOTAL_ZEROS_4x4 = {
    1: [(1, 1), (3, 3), (3, 2), (4, 3), (4, 2), (5, 3), (5, 2), (6, 3), (6, 2), (7, 3), (7, 2), (8, 3), (8, 2), (9, 3), (9, 2), (9, 1)],
    2: [(3, 7), (3, 6), (3, 5), (3, 4), (3, 3), (4, 5), (4, 4), (4, 3), (4, 2), (5, 3), (5, 2), (6, 3), (6, 2), (6, 1), (6, 0)],
    3: [(4, 5), (3, 7), (3, 6), (3, 5), (4, 4), (4, 3), (3, 4), (3, 3), (4, 2), (5, 3), (5, 2), (6, 1), (5, 1), (6, 0)],
    4: [(5, 3), (3, 7), (4, 5), (4, 4), (3, 6), (3, 5), (3, 4), (4, 3), (3, 3), (4, 2), (5, 2), (5, 1), (5, 0)],
    5: [(4, 5), (4, 4), (4, 3), (3, 7), (3, 6), (3, 5), (3, 4), (3, 3), (4, 2), (5, 1), (4, 1), (5, 0)],
    6: [(6, 1), (5, 1), (3, 7), (3, 6), (3, 5), (3, 4), (3, 3), (3, 2), (4, 1), (3, 1), (6, 0)],
    7: [(6, 1), (5, 1), (3, 5), (3, 4), (3, 3), (2, 3), (3, 2), (4, 1), (3, 1), (6, 0)],
    8: [(6, 1), (4, 1), (5, 1), (3, 3), (2, 3), (2, 2), (3, 2), (3, 1), (6, 0)],
    9: [(6, 1), (6, 0), (4, 1), (2, 3), (2, 2), (3, 1), (2, 1), (5, 1)],
    10: [(5, 1), (5, 0), (3, 1), (2, 3), (2, 2), (2, 1), (4, 1)],
    11: [(4, 0), (4, 1), (3, 1), (3, 2), (1, 1), (3, 3)],
    12: [(4, 0), (4, 1), (2, 1), (1, 1), (3, 1)],
    13: [(3, 0), (3, 1), (1, 1), (2, 1)],
    14: [(2, 0), (2, 1), (1, 1)],
    15: [(1, 0), (1, 1)],
}
TOTAL_ZEROS_CDC = {  # Table 9-9 (a), 2x2 chroma DC
    1: [(1, 1), (2, 1), (3, 1), (3, 0)],
    2: [(1, 1), (2, 1), (2, 0)],
    3: [(1, 1), (1, 0)],
}
RUN_BEFORE = {  # Table 9-10; row = min(zerosLeft, 7), entry k = run_before k
    1: [(1, 1), (1, 0)],
    2: [(1, 1), (2, 1), (2, 0)],
    3: [(2, 3), (2, 2), (2, 1), (2, 0)],
    4: [(2, 3), (2, 2), (2, 1), (3, 1), (3, 0)],
    5: [(2, 3), (2, 2), (3, 3), (3, 2), (3, 1), (3, 0)],
    6: [(2, 3), (3, 0), (3, 1), (3, 3), (3, 2), (3, 5), (3, 4)],
    7: [(3, 7), (3, 6), (3, 5), (3, 4), (3, 3), (3, 2), (3, 1), (4, 1), (5, 1), (6, 1), (7, 1), (8, 1), (9, 1), (10, 1), (11, 1)],
}

# Table 9-4, coded_block_pattern for ChromaArrayType 1: codeNum -> (Intra_4x4 / Intra_8x8, Inter)
CBP_ME = [
    (47, 0), (31, 16), (15, 1), (0, 2), (23, 4), (27, 8), (29, 32), (30, 3), (7, 5), (11, 10), (13, 12), (14, 15),
    (39, 47), (43, 7), (45, 11), (46, 13), (16, 14), (3, 6), (5, 9), (10, 31), (12, 35), (19, 37), (21, 42), (26, 44),
    (28, 33), (35, 34), (37, 36), (42, 40), (44, 39), (1, 43), (2, 45), (4, 46), (8, 17), (17, 18), (18, 20), (20, 24),
    (24, 19), (6, 21), (9, 26), (22, 28), (25, 23), (32, 27), (33, 29), (34, 30), (36, 22), (40, 25), (38, 38), (41, 41),
]

# 8.5.6: inverse zig-zag (frame) scan, index -> (x, y)
ZIGZAG_XY = [(0, 0), (1, 0), (0, 1), (0, 2), (1, 1), (2, 0), (3, 0), (2, 1), (1, 2), (0, 3), (1, 3), (2, 2), (3, 1), (3, 2), (2, 3), (3, 3)]
# 8.5.9: normAdjust4x4 by qP % 6: (positions with both coordinates even, both odd, the others)
NORM_ADJUST = [(10, 16, 13), (11, 18, 14), (13, 20, 16), (14, 23, 18), (16, 25, 20), (18, 29, 23)]
# Table 8-15: QPc as a function of qPI for qPI >= 30
QPC_HIGH = [29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39]
# Table 8-16: alpha' and beta' by indexA / indexB
ALPHA = [0] * 16 + [4, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113,
                    127, 144, 162, 182, 203, 226, 255, 255]
BETA = [0] * 16 + [2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16,
                   17, 17, 18, 18]
# Table 8-17: tC0' by indexA and bS 1..3
TC0 = [(0, 0, 0)] * 17 + [(0, 0, 1)] * 4 + [(0, 1, 1)] * 2 + [(1, 1, 1)] * 4 + [(1, 1, 2)] * 4 + [(1, 2, 3)] * 2 + [
    (2, 2, 3), (2, 2, 4), (2, 3, 4), (2, 3, 4), (3, 3, 5), (3, 4, 6), (3, 4, 6), (4, 5, 7), (4, 5, 8), (4, 6, 9), (5, 7, 10),
    (6, 8, 11), (6, 8, 13), (7, 10, 14), (8, 11, 16), (9, 12, 18), (10, 13, 20), (11, 15, 23), (13, 17, 25)]
assert len(ALPHA) == 52 and len(BETA) == 52 and len(TC0) == 52 and len(QPC_HIGH) == 22 and len(CBP_ME) == 48


def vlc_tables():
    """every VLC table of this file as {name: {symbol: (length, value)}} (for the structural tests)"""
    out = {}
    for name, tab in (("coeff_token 0<=nC<2", COEFF_TOKEN_0), ("coeff_token 2<=nC<4", COEFF_TOKEN_2),
                      ("coeff_token 4<=nC<8", COEFF_TOKEN_4), ("coeff_token 8<=nC", COEFF_TOKEN_8),
                      ("coeff_token chroma DC", COEFF_TOKEN_CDC)):
        out[name] = {(tc, t1): e for tc, row in enumerate(tab) for t1, e in enumerate(row) if e is not None}
    for k, row in TOTAL_ZEROS_4x4.items():
        out["total_zeros 4x4 tzVlcIndex %d" % k] = dict(enumerate(row))
    for k, row in TOTAL_ZEROS_CDC.items():
        out["total_zeros chroma DC tzVlcIndex %d" % k] = dict(enumerate(row))
    for k, row in RUN_BEFORE.items():
        out["run_before zerosLeft %d" % k] = dict(enumerate(row))
    return out


def _decode_map(table):
    return {(ln, val): sym for sym, (ln, val) in table.items()}


_VLC = {name: _decode_map(t) for name, t in vlc_tables().items()}


class SyntaxError264(Exception):
    pass


# ----------------------------------------------------------------------------------------
# Bit reader (7.2), Exp-Golomb (9.1)
# ----------------------------------------------------------------------------------------
class Bits:
    def __init__(self, rbsp):
        self.d = rbsp
        self.pos = 0
        self.n = len(rbsp) * 8
        # position of the rbsp_stop_one_bit: the last 1 bit of the payload
        last = len(rbsp) - 1
        while last >= 0 and rbsp[last] == 0:
            last -= 1
        if last < 0:
            raise SyntaxError264("empty RBSP")
        b = rbsp[last]
        self.stop = last * 8 + 7 - ((b & -b).bit_length() - 1)

    def u(self, n):
        v = 0
        for _ in range(n):
            if self.pos >= self.n:
                raise SyntaxError264("read past the end of the NAL unit")
            v = (v << 1) | ((self.d[self.pos >> 3] >> (7 - (self.pos & 7))) & 1)
            self.pos += 1
        return v

    def ue(self):
        z = 0
        while self.u(1) == 0:
            z += 1
            if z > 32:
                raise SyntaxError264("Exp-Golomb prefix too long")
        return (1 << z) - 1 + (self.u(z) if z else 0)

    def se(self):
        k = self.ue()
        return (k + 1) >> 1 if k & 1 else -(k >> 1)

    def more_rbsp_data(self):
        return self.pos < self.stop

    def byte_aligned(self):
        return (self.pos & 7) == 0

    def vlc(self, name):
        tab = _VLC[name]
        ln, val = 0, 0
        while ln < 16:
            val = (val << 1) | self.u(1)
            ln += 1
            s = tab.get((ln, val))
            if s is not None:
                return s
        raise SyntaxError264("no code word of table '%s' matches" % name)


def split_nals(data):
    """Annex B byte stream -> list of (nal_ref_idc, nal_unit_type, rbsp bytes)"""
    out, i, n = [], 0, len(data)
    starts = []
    while i + 2 < n:
        if data[i] == 0 and data[i + 1] == 0 and data[i + 2] == 1:
            starts.append(i + 3)
            i += 3
        else:
            i += 1
    if any(data[:starts[0] - 3] if starts else data):
        raise SyntaxError264("bytes in front of the first start code")   # (Annex B: leading_zero_8bits only)
    for k, s in enumerate(starts):
        e = starts[k + 1] - 3 if k + 1 < len(starts) else n
        while e > s and data[e - 1] == 0:  # trailing_zero_8bits / the zero_byte of the next start code
            e -= 1
        nal = data[s:e]
        if not nal:
            continue
        rbsp = bytearray()
        z = 0
        for b in nal[1:]:
            if z >= 2 and b == 3:  # emulation_prevention_three_byte
                z = 0
                continue
            if z >= 2 and b < 3:   # 7.4.1: 00 00 00, 00 00 01 and 00 00 02 never occur inside a NAL unit
                raise SyntaxError264("bytes 00 00 %02x inside a NAL unit" % b)
            rbsp.append(b)
            z = z + 1 if b == 0 else 0
        out.append(((nal[0] >> 5) & 3, nal[0] & 31, bytes(rbsp)))
    return out


def clip1(v):
    return 0 if v < 0 else 255 if v > 255 else v


def clip3(lo, hi, v):
    return lo if v < lo else hi if v > hi else v


def blk_index(bx, by):
    """4x4 block (bx, by) of a macroblock -> its index in coding order (6.4.3)"""
    return ((by >> 1) << 3) | ((bx >> 1) << 2) | ((by & 1) << 1) | (bx & 1)


BLK_XY = [None] * 16
for _by in range(4):
    for _bx in range(4):
        BLK_XY[blk_index(_bx, _by)] = (_bx, _by)


class Picture:
    def __init__(self, w, h):
        self.w, self.h = w, h
        self.y = [bytearray(w) for _ in range(h)]
        self.cb = [bytearray(w // 2) for _ in range(h // 2)]
        self.cr = [bytearray(w // 2) for _ in range(h // 2)]

    def luma_bytes(self):
        return b"".join(bytes(r) for r in self.y)

    def nv12_chroma_bytes(self):
        out = bytearray()
        for rb, rr in zip(self.cb, self.cr):
            row = bytearray(len(rb) * 2)
            row[0::2] = rb
            row[1::2] = rr
            out += row
        return bytes(out)


class MB:
    __slots__ = ("avail", "slice", "intra", "i4", "modes", "skip", "mv", "ref", "nz", "nzc", "qp", "qpc", "pcm")

    def __init__(self):
        self.avail = False   # decoded already (in this picture)
        self.slice = -1
        self.intra = False
        self.i4 = False      # Intra4x4 (its block modes are in .modes)
        self.modes = [2] * 16
        self.skip = False
        self.mv = (0, 0)     # quarter-sample units
        self.ref = -1        # reference index, -1 = none (intra)
        self.nz = [0] * 16   # TotalCoeff per luma 4x4 block, index = by * 4 + bx
        self.nzc = [[0] * 4, [0] * 4]
        self.qp = 0
        self.qpc = 0
        self.pcm = False


# Table A-1 (level limits), the columns this decoder checks: level_idc -> (MaxMBPS macroblocks/s, MaxFS macroblocks,
# MaxVmvR: vertical vector components lie in [-MaxVmvR, MaxVmvR - 0.25] luma samples). Level 1b shares 11's code
# in the baseline profile with constraint_set3_flag; these streams never use it.
LEVEL_LIMITS = {
    10: (1485, 99, 64), 11: (3000, 396, 128), 12: (6000, 396, 128), 13: (11880, 396, 128),
    20: (11880, 396, 128), 21: (19800, 792, 256), 22: (20250, 1620, 256),
    30: (40500, 1620, 256), 31: (108000, 3600, 512), 32: (216000, 5120, 512),
    40: (245760, 8192, 512), 41: (245760, 8192, 512), 42: (522240, 8704, 512),
    50: (589824, 22080, 512), 51: (983040, 36864, 512), 52: (2073600, 36864, 512),
}
MAX_MB_BITS = 3200   # A.3.1: bits of macroblock_layer() of any macroblock (I_PCM: 3072 + its header fits)


class Decoder:
    def __init__(self):
        self.max_abs_mv = [0, 0]   # largest |mvx|, |mvy| seen, quarter samples
        self.sps = None
        self.pps = None
        self.cur = None
        self.ref_pic = None
        self.mbs = None
        self.cover = Counter()
        self.pictures = 0
        self.slice_no = 0
        self.deblock_idc = []      # per slice: (idc, alpha offset, beta offset)
        self.pending = False       # a picture is being decoded (slices seen, not finished)
        self.max_mb_bits = 0       # the largest macroblock_layer() seen, in bits
        self.next_mb = 0

    # ---------------- parameter sets ----------------
    def _parse_sps(self, b):
        s = {"profile_idc": b.u(8), "constraints": b.u(8), "level_idc": b.u(8), "id": b.ue()}
        if s["profile_idc"] in (100, 110, 122, 244, 44, 83, 86, 118, 128, 138, 139, 134, 135):
            raise SyntaxError264("high profiles are outside this decoder")
        s["log2_max_frame_num"] = b.ue() + 4
        s["poc_type"] = b.ue()
        if s["poc_type"] == 0:
            s["log2_max_poc_lsb"] = b.ue() + 4
        elif s["poc_type"] == 1:
            raise SyntaxError264("pic_order_cnt_type 1 is outside this decoder")
        s["max_num_ref_frames"] = b.ue()
        s["gaps_allowed"] = b.u(1)
        s["mb_w"] = b.ue() + 1
        s["mb_h"] = b.ue() + 1
        if not b.u(1):
            raise SyntaxError264("frame_mbs_only_flag = 0 is outside this decoder")
        s["direct_8x8"] = b.u(1)
        s["crop"] = (0, 0, 0, 0)
        if b.u(1):
            s["crop"] = (b.ue(), b.ue(), b.ue(), b.ue())  # left, right, top, bottom, in units of 2 samples
        s["vui"] = b.u(1)
        s["fps"] = None
        if s["vui"]:
            self._parse_vui(b, s)
        self.sps = s
        self.cover["sps"] += 1

    def _parse_vui(self, b, s):
        """E.1.1 vui_parameters(); what is kept: the frame rate and the bitstream restrictions"""
        if b.u(1):                       # aspect_ratio_info_present_flag
            if b.u(8) == 255:            # Extended_SAR
                b.u(16), b.u(16)
        if b.u(1):                       # overscan_info_present_flag
            b.u(1)
        if b.u(1):                       # video_signal_type_present_flag
            b.u(3), b.u(1)
            if b.u(1):                   # colour_description_present_flag
                b.u(8), b.u(8), b.u(8)
        if b.u(1):                       # chroma_loc_info_present_flag
            b.ue(), b.ue()
        if b.u(1):                       # timing_info_present_flag
            num_units_in_tick, time_scale = b.u(32), b.u(32)
            s["fixed_frame_rate"] = b.u(1)
            if num_units_in_tick == 0 or time_scale == 0:
                raise SyntaxError264("timing_info with a zero")
            s["fps"] = (time_scale, 2 * num_units_in_tick)   # a frame is two ticks (E.2.1)
            self.cover["vui_timing"] += 1

        def hrd():
            cpb_cnt = b.ue() + 1
            b.u(4), b.u(4)
            for _ in range(cpb_cnt):
                b.ue(), b.ue(), b.u(1)
            b.u(5), b.u(5), b.u(5), b.u(5)
        nal_hrd = b.u(1)
        if nal_hrd:
            hrd()
        vcl_hrd = b.u(1)
        if vcl_hrd:
            hrd()
        if nal_hrd or vcl_hrd:
            b.u(1)                       # low_delay_hrd_flag
        b.u(1)                           # pic_struct_present_flag
        if b.u(1):                       # bitstream_restriction_flag
            r = {"mv_over_pic_boundaries": b.u(1), "max_bytes_per_pic_denom": b.ue(), "max_bits_per_mb_denom": b.ue(),
                 "log2_max_mv_length_horizontal": b.ue(), "log2_max_mv_length_vertical": b.ue(),
                 "max_num_reorder_frames": b.ue(), "max_dec_frame_buffering": b.ue()}
            if r["max_bits_per_mb_denom"] > 16 or r["log2_max_mv_length_horizontal"] > 16 or r["log2_max_mv_length_vertical"] > 16:
                raise SyntaxError264("bitstream_restriction out of range")
            s["restriction"] = r
            self.cover["vui_restriction"] += 1

    def level_violations(self):
        """A.3.1 for the stream decoded so far, as a list of sentences (empty: within its level). Checked: frame size
        and shape (MaxFS), macroblock rate at the VUI frame rate (MaxMBPS), bits per macroblock (3200), the vertical
        and horizontal vector range, and the VUI's own promises (log2_max_mv_length_*, max_bits_per_mb_denom)."""
        s, out = self.sps, []
        if s is None:
            return ["no sequence parameter set"]
        lim = LEVEL_LIMITS.get(s["level_idc"])
        if lim is None:
            return ["level_idc %d is not in Table A-1" % s["level_idc"]]
        mbps, fs, vmvr = lim
        n = s["mb_w"] * s["mb_h"]
        if n > fs:
            out.append("PicSizeInMbs %d > MaxFS %d" % (n, fs))
        if s["mb_w"] ** 2 > 8 * fs or s["mb_h"] ** 2 > 8 * fs:
            out.append("picture %dx%d macroblocks is wider or taller than sqrt(8 MaxFS)" % (s["mb_w"], s["mb_h"]))
        if s["fps"] and n * s["fps"][0] > mbps * s["fps"][1]:
            out.append("%d macroblocks at %d/%d frames/s > MaxMBPS %d" % (n, s["fps"][0], s["fps"][1], mbps))
        if self.max_mb_bits > MAX_MB_BITS:
            out.append("a macroblock of %d bits > %d" % (self.max_mb_bits, MAX_MB_BITS))
        if self.max_abs_mv[1] > 4 * vmvr:   # (+MaxVmvR itself is out: the range ends at MaxVmvR - 0.25; |v| == 4 MaxVmvR only passes for v < 0)
            out.append("vertical vector component %d/4 outside +-%d" % (self.max_abs_mv[1], vmvr))
        if self.max_abs_mv[0] > 4 * 2048:
            out.append("horizontal vector component %d/4 outside +-2048" % self.max_abs_mv[0])
        r = s.get("restriction")
        if r:
            if self.max_abs_mv[0] >= 1 << r["log2_max_mv_length_horizontal"] or self.max_abs_mv[1] >= 1 << r["log2_max_mv_length_vertical"]:
                out.append("a vector of (%d, %d)/4 breaks the VUI's log2_max_mv_length" % tuple(self.max_abs_mv))
            if r["max_bits_per_mb_denom"] and self.max_mb_bits > (128 + 3072) // r["max_bits_per_mb_denom"]:
                out.append("a macroblock of %d bits breaks the VUI's max_bits_per_mb_denom" % self.max_mb_bits)
        return out

    def _parse_pps(self, b):
        p = {"id": b.ue(), "sps_id": b.ue(), "cabac": b.u(1), "bottom_field_pic_order": b.u(1)}
        if p["cabac"]:
            raise SyntaxError264("CABAC is outside this decoder")
        if b.ue() != 0:
            raise SyntaxError264("slice groups are outside this decoder")
        p["num_ref_idx_l0"] = b.ue() + 1
        p["num_ref_idx_l1"] = b.ue() + 1
        if b.u(1) or b.u(2):
            raise SyntaxError264("weighted prediction is outside this decoder")
        p["init_qp"] = 26 + b.se()
        p["init_qs"] = 26 + b.se()
        p["chroma_qp_offset"] = b.se()
        p["deblock_control"] = b.u(1)
        p["constrained_intra"] = b.u(1)
        p["redundant_pic_cnt"] = b.u(1)
        self.pps = p
        self.cover["pps"] += 1

    # ---------------- entry point ----------------
    def decode(self, data):
        """feed one or more NAL units (Annex B); returns the pictures completed by this call"""
        done = []
        for ref_idc, typ, rbsp in split_nals(bytes(data)):
            if typ == 7:
                self._parse_sps(Bits(rbsp))
            elif typ == 8:
                self._parse_pps(Bits(rbsp))
            elif typ in (1, 5):
                pic = self._slice(Bits(rbsp), ref_idc, typ)
                if pic is not None:
                    done.append(pic)
            elif typ == 9:
                self.cover["aud"] += 1
            else:
                self.cover["nal_other"] += 1
        return done

    # ---------------- slice layer (7.3.3, 7.3.4) ----------------
    def _slice(self, b, ref_idc, typ):
        if self.sps is None or self.pps is None:
            raise SyntaxError264("slice before SPS / PPS")
        sps, pps = self.sps, self.pps
        first_mb = b.ue()
        st = b.ue()
        slice_type = st % 5
        if slice_type not in (0, 2):
            raise SyntaxError264("slice_type %d is outside this decoder" % st)
        is_p = slice_type == 0
        if b.ue() != pps["id"]:
            raise SyntaxError264("unknown pic_parameter_set_id")
        frame_num = b.u(sps["log2_max_frame_num"])
        idr = typ == 5
        if idr:
            self.idr_pic_id = b.ue()
            if is_p:
                raise SyntaxError264("P slice in an IDR picture")
        if sps["poc_type"] == 0:
            b.u(sps["log2_max_poc_lsb"])
        if pps["redundant_pic_cnt"]:
            b.ue()
        if is_p:
            if b.u(1):  # num_ref_idx_active_override_flag
                if b.ue() != 0:
                    raise SyntaxError264("more than one reference picture is outside this decoder")
            elif pps["num_ref_idx_l0"] != 1:
                raise SyntaxError264("more than one reference picture is outside this decoder")
            if b.u(1):
                raise SyntaxError264("reference picture list modification is outside this decoder")
        if ref_idc:
            if idr:
                b.u(1)
                if b.u(1):
                    raise SyntaxError264("long-term references are outside this decoder")
            elif b.u(1):
                raise SyntaxError264("adaptive reference picture marking is outside this decoder")
        qp = pps["init_qp"] + b.se()
        idc, a_off, b_off = 0, 0, 0
        if pps["deblock_control"]:
            idc = b.ue()
            if idc != 1:
                a_off, b_off = 2 * b.se(), 2 * b.se()
        self.cover["slice_I" if not is_p else "slice_P"] += 1
        self.cover["deblock_idc_%d" % idc] += 1

        mb_w, mb_h = sps["mb_w"], sps["mb_h"]
        if first_mb == 0 or not self.pending:
            # a new picture (the streams here always start pictures with first_mb_in_slice = 0)
            if first_mb != 0:
                raise SyntaxError264("picture does not start at macroblock 0")
            if self.pending:
                raise SyntaxError264("a new picture starts before the previous one is complete")
            self.cur = Picture(mb_w * 16, mb_h * 16)
            self.mbs = [MB() for _ in range(mb_w * mb_h)]
            self.slice_no = 0
            self.deblock_idc = []
            self.pending = True
            self.next_mb = 0
            self.frame_num = frame_num
            self.is_idr = idr
            if is_p and self.ref_pic is None:
                raise SyntaxError264("P picture without a reference picture")
        else:
            if first_mb != self.next_mb:
                raise SyntaxError264("slice starts at macroblock %d, expected %d" % (first_mb, self.next_mb))
            if frame_num != self.frame_num:
                raise SyntaxError264("frame_num changes inside a picture")
            self.slice_no += 1
        self.deblock_idc.append((idc, a_off, b_off))
        self.qp = qp

        # ---- slice_data (7.3.4), CAVLC ----
        addr = first_mb
        n_mb = mb_w * mb_h
        more = True
        while more:
            if is_p:
                run = b.ue()
                self.cover["mb_skip_run_%s" % ("0" if run == 0 else "1" if run == 1 else "n")] += 1
                for _ in range(run):
                    if addr >= n_mb:
                        raise SyntaxError264("mb_skip_run runs past the picture")
                    self._p_skip(addr)
                    addr += 1
                if run > 0:        # 7.3.4: moreDataFlag is only looked at again after a run of skipped macroblocks
                    more = b.more_rbsp_data()
                if not more:
                    break
            if addr >= n_mb:
                raise SyntaxError264("macroblock data past the end of the picture")
            pos0 = b.pos
            self._macroblock(b, addr, is_p)
            self.max_mb_bits = max(self.max_mb_bits, b.pos - pos0)   # A.3.1: macroblock_layer() <= 3200 bits
            if b.pos > b.stop:
                raise SyntaxError264("macroblock data runs into rbsp_trailing_bits")
            addr += 1
            more = b.more_rbsp_data()
        self.next_mb = addr
        if addr == n_mb:
            return self._finish_picture()
        return None

    def _finish_picture(self):
        self._deblock_picture()
        self.pending = False
        self.ref_pic = self.cur
        self.pictures += 1
        return self.cur

    # ---------------- neighbours (6.4.8 .. 6.4.11) ----------------
    def _mb_at(self, mx, my):
        """the macroblock at (mx, my) if it is available to the current slice, else None"""
        s = self.sps
        if mx < 0 or my < 0 or mx >= s["mb_w"] or my >= s["mb_h"]:
            return None
        m = self.mbs[my * s["mb_w"] + mx]
        return m if m.avail and m.slice == self.slice_no else None

    def _mb_at_intra(self, mx, my):
        """the neighbour as intra prediction sees it: with constrained_intra_pred_flag a macroblock coded
        in Inter prediction mode is "not available for Intra prediction" (8.3.1.2, 8.3.3, 8.3.4) and sets
        dcPredModePredictedFlag (8.3.1.1)"""
        m = self._mb_at(mx, my)
        if m is not None and self.pps["constrained_intra"] and not m.intra:
            self.cover["constrained_intra_neighbour_hidden"] += 1
            return None
        return m

    # ---------------- macroblock layer (7.3.5) ----------------
    def _macroblock(self, b, addr, is_p):
        sps, pps = self.sps, self.pps
        mb_w = sps["mb_w"]
        mx, my = addr % mb_w, addr // mb_w
        m = self.mbs[addr]
        m.slice = self.slice_no
        mb_type = b.ue()
        if is_p:
            if mb_type >= 5:
                mb_type -= 5
                intra = True
                self.cover["intra_mb_in_P_slice"] += 1
            else:
                intra = False
                if mb_type != 0:
                    raise SyntaxError264("P macroblock type %d (partitions) is outside this decoder" % mb_type)
        else:
            intra = True
        m.intra = intra
        m.skip = False
        m.pcm = False
        if intra and mb_type == 25:
            return self._i_pcm(b, m, mx, my)
        i16 = intra and mb_type >= 1
        m.i4 = intra and mb_type == 0
        m.ref = -1 if intra else 0
        m.mv = (0, 0)
        A, B = self._mb_at(mx - 1, my), self._mb_at(mx, my - 1)
        if intra and not i16:
            # mb_pred: Intra4x4 prediction modes (8.3.1.1)
            for blk in range(16):
                bx, by = BLK_XY[blk]
                pred = self._pred_i4_mode(m, self._mb_at_intra(mx - 1, my), self._mb_at_intra(mx, my - 1), bx, by)
                if b.u(1):
                    mode = pred
                    self.cover["i4_mode_predicted"] += 1
                else:
                    rem = b.u(3)
                    mode = rem if rem < pred else rem + 1
                    self.cover["i4_mode_rem"] += 1
                m.modes[blk] = mode
        if intra:
            chroma_mode = b.ue()
            if chroma_mode > 3:
                raise SyntaxError264("intra_chroma_pred_mode %d" % chroma_mode)
        else:
            # P_L0_16x16: ref_idx absent (one reference), mvd_l0
            mvdx, mvdy = b.se(), b.se()
            px, py = self._mv_pred_16x16(mx, my)
            m.mv = (px + mvdx, py + mvdy)
            self.max_abs_mv = [max(self.max_abs_mv[0], abs(m.mv[0])), max(self.max_abs_mv[1], abs(m.mv[1]))]
            self.cover["mvd_zero" if (mvdx, mvdy) == (0, 0) else "mvd_nonzero"] += 1
        if i16:
            i16_mode = (mb_type - 1) & 3
            cbp_c = ((mb_type - 1) >> 2) % 3
            cbp_l = 15 if mb_type >= 13 else 0
            cbp = cbp_l | (cbp_c << 4)
        else:
            code = b.ue()
            if code > 47:
                raise SyntaxError264("coded_block_pattern code %d" % code)
            cbp = CBP_ME[code][0 if intra else 1]
            self.cover["cbp_me_%s_%d" % ("intra" if intra else "inter", code)] += 1
        if cbp or i16:
            dqp = b.se()
            if dqp < -26 or dqp > 25:
                raise SyntaxError264("mb_qp_delta %d" % dqp)
            self.qp = (self.qp + dqp + 52) % 52
            if dqp:
                self.cover["mb_qp_delta_nonzero"] += 1
        m.qp = self.qp
        m.qpc = self._qpc(self.qp)

        # ---- residual (7.3.5.3) ----
        m.nz = [0] * 16
        m.nzc = [[0] * 4, [0] * 4]
        luma = [[0] * 16 for _ in range(16)]   # per block (coding order): coefficients in zig-zag order
        dc16 = None
        if i16:
            dc16 = self._residual_block(b, self._nc_luma(m, A, B, 0, 0), 16)[0]
        for blk in range(16):
            bx, by = BLK_XY[blk]
            if cbp & (1 << (blk >> 2)):
                nc = self._nc_luma(m, A, B, bx, by)
                if i16:
                    c, tc = self._residual_block(b, nc, 15)
                    luma[blk][1:] = c[:15]
                else:
                    c, tc = self._residual_block(b, nc, 16)
                    luma[blk] = c
                m.nz[by * 4 + bx] = tc
        cdc = [[0] * 4, [0] * 4]
        cac = [[[0] * 16 for _ in range(4)] for _ in range(2)]
        if cbp & 0x30:
            for c in range(2):
                cdc[c] = self._residual_block(b, -1, 4)[0][:4]
        if cbp & 0x20:
            for c in range(2):
                for blk in range(4):
                    nc = self._nc_chroma(m, A, B, c, blk & 1, blk >> 1)
                    co, tc = self._residual_block(b, nc, 15)
                    cac[c][blk][1:] = co[:15]
                    m.nzc[c][blk] = tc
        m.avail = True

        # ---- prediction + reconstruction ----
        pic = self.cur
        x0, y0 = mx * 16, my * 16
        qp, qpc = m.qp, m.qpc
        if not intra:
            self._inter_pred(m, mx, my)
            self.cover["mb_P_L0_16x16"] += 1
            for blk in range(16):
                bx, by = BLK_XY[blk]
                if m.nz[by * 4 + bx]:
                    self._add_residual(pic.y, x0 + 4 * bx, y0 + 4 * by, self._dequant4x4(luma[blk], qp, False))
        elif i16:
            self._pred_i16(mx, my, i16_mode)
            self.cover["mb_I16x16"] += 1
            self.cover["i16_mode_%d" % i16_mode] += 1
            dcs = self._i16_dc(dc16, qp)
            for blk in range(16):
                bx, by = BLK_XY[blk]
                res = self._dequant4x4(luma[blk], qp, True)
                res[0] = dcs[by * 4 + bx]
                self._add_residual(pic.y, x0 + 4 * bx, y0 + 4 * by, res)
        else:
            self.cover["mb_I4x4"] += 1
            for blk in range(16):
                bx, by = BLK_XY[blk]
                self._pred_i4(mx, my, bx, by, m.modes[blk])
                self.cover["i4_mode_%d" % m.modes[blk]] += 1
                if m.nz[by * 4 + bx]:
                    self._add_residual(pic.y, x0 + 4 * bx, y0 + 4 * by, self._dequant4x4(luma[blk], qp, False))
        if intra:
            self._pred_chroma(mx, my, chroma_mode)
            self.cover["chroma_mode_%d" % chroma_mode] += 1
        for c, plane in enumerate((pic.cb, pic.cr)):
            if not (cbp & 0x30):
                continue
            f = cdc[c]
            # 8.5.11.1 / 8.5.11.2: 2x2 transform of the chroma DC levels, then scaling
            h = [f[0] + f[1] + f[2] + f[3], f[0] - f[1] + f[2] - f[3], f[0] + f[1] - f[2] - f[3], f[0] - f[1] - f[2] + f[3]]
            ls = NORM_ADJUST[qpc % 6][0] * 16
            dcc = [((v * ls) << (qpc // 6)) >> 5 for v in h]
            for blk in range(4):
                res = self._dequant4x4(cac[c][blk], qpc, True)
                res[0] = dcc[blk]
                self._add_residual(plane, mx * 8 + 4 * (blk & 1), my * 8 + 4 * (blk >> 1), res)

    def _qpc(self, qp):
        qpi = clip3(0, 51, qp + self.pps["chroma_qp_offset"])
        return qpi if qpi < 30 else QPC_HIGH[qpi - 30]

    def _i_pcm(self, b, m, mx, my):
        while not b.byte_aligned():
            if b.u(1):
                raise SyntaxError264("pcm_alignment_zero_bit is 1")
        pic = self.cur
        for y in range(16):
            for x in range(16):
                pic.y[my * 16 + y][mx * 16 + x] = b.u(8)
        for plane in (pic.cb, pic.cr):
            for y in range(8):
                for x in range(8):
                    plane[my * 8 + y][mx * 8 + x] = b.u(8)
        m.intra, m.i4, m.pcm, m.ref, m.mv = True, False, True, -1, (0, 0)
        m.nz = [16] * 16
        m.nzc = [[16] * 4, [16] * 4]
        m.qp = 0             # qPp of an I_PCM macroblock is 0 in the deblocking filter
        m.qpc = self._qpc(0)
        m.avail = True
        self.cover["mb_I_PCM"] += 1

    # ---------------- CAVLC residual block (7.3.5.3.2, 9.2) ----------------
    def _nc_from(self, na, nb):
        if na is not None and nb is not None:
            return (na + nb + 1) >> 1
        if na is not None:
            return na
        if nb is not None:
            return nb
        return 0

    def _nc_luma(self, m, A, B, bx, by):
        na = m.nz[by * 4 + bx - 1] if bx > 0 else (A.nz[by * 4 + 3] if A is not None else None)
        nb = m.nz[(by - 1) * 4 + bx] if by > 0 else (B.nz[12 + bx] if B is not None else None)
        return self._nc_from(na, nb)

    def _nc_chroma(self, m, A, B, c, bx, by):
        na = m.nzc[c][by * 2] if bx > 0 else (A.nzc[c][by * 2 + 1] if A is not None else None)
        nb = m.nzc[c][bx] if by > 0 else (B.nzc[c][2 + bx] if B is not None else None)
        return self._nc_from(na, nb)

    def _residual_block(self, b, nc, max_coeff):
        """returns (coefficients in scan order, padded to 16, and TotalCoeff)"""
        if nc < 0:
            name = "coeff_token chroma DC"
        elif nc < 2:
            name = "coeff_token 0<=nC<2"
        elif nc < 4:
            name = "coeff_token 2<=nC<4"
        elif nc < 8:
            name = "coeff_token 4<=nC<8"
        else:
            name = "coeff_token 8<=nC"
        total, t1 = b.vlc(name)
        self.cover["coeff_token/%s" % name] += 1
        coef = [0] * 16
        if total == 0:
            return coef, 0
        if total > max_coeff:
            raise SyntaxError264("TotalCoeff %d in a block of %d" % (total, max_coeff))
        level = [0] * total
        suffix_len = 1 if (total > 10 and t1 < 3) else 0
        for i in range(total):
            if i < t1:
                level[i] = -1 if b.u(1) else 1
                continue
            prefix = 0
            while b.u(1) == 0:
                prefix += 1
                if prefix > 15:    # (9.2.2.1: more than 15 only in the High profiles, which are outside this decoder)
                    raise SyntaxError264("level_prefix too long")
            code = min(15, prefix) << suffix_len
            if suffix_len > 0 or prefix >= 14:
                size = 4 if (prefix == 14 and suffix_len == 0) else (prefix - 3 if prefix >= 15 else suffix_len)
                if size:
                    code += b.u(size)
            if prefix >= 15 and suffix_len == 0:
                code += 15
            if prefix >= 16:
                code += (1 << (prefix - 3)) - 4096
            if i == t1 and t1 < 3:
                code += 2
            level[i] = (code + 2) >> 1 if code % 2 == 0 else (-code - 1) >> 1
            self.cover["level_prefix_%s" % ("lt14" if prefix < 14 else str(prefix) if prefix < 16 else "ge16")] += 1
            self.cover["suffix_length_%d" % suffix_len] += 1
            if suffix_len == 0:
                suffix_len = 1
            if abs(level[i]) > (3 << (suffix_len - 1)) and suffix_len < 6:
                suffix_len += 1
        if total < max_coeff:
            if max_coeff == 4:
                zeros = b.vlc("total_zeros chroma DC tzVlcIndex %d" % total)
                self.cover["total_zeros_cdc_row_%d" % total] += 1
            else:
                zeros = b.vlc("total_zeros 4x4 tzVlcIndex %d" % total)
                self.cover["total_zeros_row_%d" % total] += 1
            if total + zeros > max_coeff:
                raise SyntaxError264("total_zeros %d with TotalCoeff %d in a block of %d" % (zeros, total, max_coeff))
        else:
            zeros = 0
        run = [0] * total
        left = zeros
        for i in range(total - 1):
            if left > 0:
                row = min(left, 7)
                run[i] = b.vlc("run_before zerosLeft %d" % row)
                self.cover["run_before_row_%d" % row] += 1
                if run[i] > left:
                    raise SyntaxError264("run_before exceeds zerosLeft")
                left -= run[i]
        run[total - 1] = left
        pos = -1
        for i in range(total - 1, -1, -1):
            pos += run[i] + 1
            coef[pos] = level[i]
        return coef, total

    # ---------------- scaling and transform (8.5.9 .. 8.5.12) ----------------
    def _dequant4x4(self, zz, qp, has_dc):
        """levels in zig-zag order -> 16 scaled coefficients d[4*y + x]; with has_dc the DC position is left 0"""
        na = NORM_ADJUST[qp % 6]
        sh = qp // 6
        d = [0] * 16
        for k in range(1 if has_dc else 0, 16):
            v = zz[k]
            if not v:
                continue
            x, y = ZIGZAG_XY[k]
            ls = 16 * (na[0] if (x % 2 == 0 and y % 2 == 0) else na[1] if (x % 2 == 1 and y % 2 == 1) else na[2])
            if qp >= 24:
                d[4 * y + x] = (v * ls) << (sh - 4)
            else:
                d[4 * y + x] = (v * ls + (1 << (3 - sh))) >> (4 - sh)
        return d

    def _i16_dc(self, zz, qp):
        """Intra16x16 DC levels (zig-zag) -> the 16 scaled DC coefficients, index = by * 4 + bx"""
        c = [[0] * 4 for _ in range(4)]
        for k in range(16):
            x, y = ZIGZAG_XY[k]
            c[y][x] = zz[k]

        def had(v):
            a, b2, c2, d = v
            return [a + b2 + c2 + d, a + b2 - c2 - d, a - b2 - c2 + d, a - b2 + c2 - d]
        rows = [had(r) for r in c]
        f = [[0] * 4 for _ in range(4)]
        for x in range(4):
            col = had([rows[y][x] for y in range(4)])
            for y in range(4):
                f[y][x] = col[y]
        ls = 16 * NORM_ADJUST[qp % 6][0]
        out = [0] * 16
        for y in range(4):
            for x in range(4):
                v = f[y][x] * ls
                out[y * 4 + x] = v << (qp // 6 - 6) if qp >= 36 else (v + (1 << (5 - qp // 6))) >> (6 - qp // 6)
        return out

    def _add_residual(self, plane, x0, y0, d):
        """8.5.12.2 inverse transform of d[4*y + x], then add to the prediction in place"""
        t = [0] * 16
        for y in range(4):
            d0, d1, d2, d3 = d[4 * y], d[4 * y + 1], d[4 * y + 2], d[4 * y + 3]
            e0, e1, e2, e3 = d0 + d2, d0 - d2, (d1 >> 1) - d3, d1 + (d3 >> 1)
            t[4 * y], t[4 * y + 1], t[4 * y + 2], t[4 * y + 3] = e0 + e3, e1 + e2, e1 - e2, e0 - e3
        for x in range(4):
            f0, f1, f2, f3 = t[x], t[4 + x], t[8 + x], t[12 + x]
            g0, g1, g2, g3 = f0 + f2, f0 - f2, (f1 >> 1) - f3, f1 + (f3 >> 1)
            col = (g0 + g3, g1 + g2, g1 - g2, g0 - g3)
            for y in range(4):
                row = plane[y0 + y]
                row[x0 + x] = clip1(row[x0 + x] + ((col[y] + 32) >> 6))

    # ---------------- intra prediction (8.3) ----------------
    def _pred_i4_mode(self, m, A, B, bx, by):
        def mode_of(nb, nbx, nby):
            if nb is None:
                return None          # not available: dcPredModePredictedFlag
            if not nb.i4:
                return 2             # Intra16x16, I_PCM or inter (constrained_intra_pred_flag = 0)
            return nb.modes[blk_index(nbx, nby)]
        a = m.modes[blk_index(bx - 1, by)] if bx > 0 else mode_of(A, 3, by)
        b2 = m.modes[blk_index(bx, by - 1)] if by > 0 else mode_of(B, bx, 3)
        if a is None or b2 is None:
            return 2
        return min(a, b2)

    def _pred_i4(self, mx, my, bx, by, mode):
        pic = self.cur.y
        x0, y0 = mx * 16 + 4 * bx, my * 16 + 4 * by
        has_l = bx > 0 or self._mb_at_intra(mx - 1, my) is not None
        has_t = by > 0 or self._mb_at_intra(mx, my - 1) is not None
        if by > 0:
            has_tr = bx < 3 and blk_index(bx + 1, by - 1) < blk_index(bx, by)
        elif bx < 3:
            has_tr = has_t
        else:
            has_tr = self._mb_at_intra(mx + 1, my - 1) is not None
        if by > 0 and bx > 0:
            has_tl = True
        elif by > 0:
            has_tl = has_l
        elif bx > 0:
            has_tl = has_t
        else:
            has_tl = self._mb_at_intra(mx - 1, my - 1) is not None
        T = [pic[y0 - 1][x0 + i] for i in range(4)] if has_t else None
        if has_t:
            T += [pic[y0 - 1][x0 + 4 + i] for i in range(4)] if has_tr else [T[3]] * 4
            if not has_tr:
                self.cover["i4_top_right_substituted"] += 1
        L = [pic[y0 + i][x0 - 1] for i in range(4)] if has_l else None
        M = pic[y0 - 1][x0 - 1] if has_tl else None
        need = {0: (1, 0, 0), 1: (0, 1, 0), 2: (0, 0, 0), 3: (1, 0, 0), 4: (1, 1, 1), 5: (1, 1, 1), 6: (1, 1, 1), 7: (1, 0, 0), 8: (0, 1, 0)}[mode]
        if (need[0] and not has_t) or (need[1] and not has_l) or (need[2] and not has_tl):
            raise SyntaxError264("Intra4x4 mode %d needs neighbours that are not available" % mode)

        def p(x, y):  # the standard's p[x, y] with x, y = -1 on the borders
            if y < 0:
                return M if x < 0 else T[x]
            return L[y]
        out = [[0] * 4 for _ in range(4)]
        for y in range(4):
            for x in range(4):
                if mode == 0:
                    v = T[x]
                elif mode == 1:
                    v = L[y]
                elif mode == 2:
                    if has_t and has_l:
                        v = (sum(T[:4]) + sum(L) + 4) >> 3
                    elif has_l:
                        v = (sum(L) + 2) >> 2
                    elif has_t:
                        v = (sum(T[:4]) + 2) >> 2
                    else:
                        v = 128
                elif mode == 3:
                    v = (T[6] + 3 * T[7] + 2) >> 2 if x == 3 and y == 3 else (T[x + y] + 2 * T[x + y + 1] + T[x + y + 2] + 2) >> 2
                elif mode == 4:
                    if x > y:
                        v = (p(x - y - 2, -1) + 2 * p(x - y - 1, -1) + p(x - y, -1) + 2) >> 2
                    elif x < y:
                        v = (p(-1, y - x - 2) + 2 * p(-1, y - x - 1) + p(-1, y - x) + 2) >> 2
                    else:
                        v = (p(0, -1) + 2 * p(-1, -1) + p(-1, 0) + 2) >> 2
                elif mode == 5:
                    z = 2 * x - y
                    if z >= 0 and z % 2 == 0:
                        v = (p(x - (y >> 1) - 1, -1) + p(x - (y >> 1), -1) + 1) >> 1
                    elif z >= 0:
                        v = (p(x - (y >> 1) - 2, -1) + 2 * p(x - (y >> 1) - 1, -1) + p(x - (y >> 1), -1) + 2) >> 2
                    elif z == -1:
                        v = (p(-1, 0) + 2 * p(-1, -1) + p(0, -1) + 2) >> 2
                    else:
                        v = (p(-1, y - 1) + 2 * p(-1, y - 2) + p(-1, y - 3) + 2) >> 2
                elif mode == 6:
                    z = 2 * y - x
                    if z >= 0 and z % 2 == 0:
                        v = (p(-1, y - (x >> 1) - 1) + p(-1, y - (x >> 1)) + 1) >> 1
                    elif z >= 0:
                        v = (p(-1, y - (x >> 1) - 2) + 2 * p(-1, y - (x >> 1) - 1) + p(-1, y - (x >> 1)) + 2) >> 2
                    elif z == -1:
                        v = (p(-1, 0) + 2 * p(-1, -1) + p(0, -1) + 2) >> 2
                    else:
                        v = (p(x - 1, -1) + 2 * p(x - 2, -1) + p(x - 3, -1) + 2) >> 2
                elif mode == 7:
                    i = x + (y >> 1)
                    v = (T[i] + T[i + 1] + 1) >> 1 if y % 2 == 0 else (T[i] + 2 * T[i + 1] + T[i + 2] + 2) >> 2
                else:
                    z = x + 2 * y
                    i = y + (x >> 1)
                    if z > 5:
                        v = L[3]
                    elif z == 5:
                        v = (L[2] + 3 * L[3] + 2) >> 2
                    elif z % 2 == 0:
                        v = (L[i] + L[i + 1] + 1) >> 1
                    else:
                        v = (L[i] + 2 * L[i + 1] + L[i + 2] + 2) >> 2
                out[y][x] = v
        for y in range(4):
            for x in range(4):
                pic[y0 + y][x0 + x] = out[y][x]

    def _pred_i16(self, mx, my, mode):
        pic = self.cur.y
        x0, y0 = mx * 16, my * 16
        has_l = self._mb_at_intra(mx - 1, my) is not None
        has_t = self._mb_at_intra(mx, my - 1) is not None
        T = [pic[y0 - 1][x0 + i] for i in range(16)] if has_t else None
        L = [pic[y0 + i][x0 - 1] for i in range(16)] if has_l else None
        if (mode == 0 and not has_t) or (mode == 1 and not has_l) or (mode == 3 and not (has_t and has_l and self._mb_at_intra(mx - 1, my - 1) is not None)):
            raise SyntaxError264("Intra16x16 mode %d needs neighbours that are not available" % mode)
        if mode == 2:
            if has_t and has_l:
                dc = (sum(T) + sum(L) + 16) >> 5
            elif has_l:
                dc = (sum(L) + 8) >> 4
            elif has_t:
                dc = (sum(T) + 8) >> 4
            else:
                dc = 128
        if mode == 3:
            M = pic[y0 - 1][x0 - 1]
            tt = lambda i: M if i < 0 else T[i]
            ll = lambda i: M if i < 0 else L[i]
            H = sum((k + 1) * (tt(8 + k) - tt(6 - k)) for k in range(8))
            V = sum((k + 1) * (ll(8 + k) - ll(6 - k)) for k in range(8))
            a = 16 * (L[15] + T[15])
            bb, cc = (5 * H + 32) >> 6, (5 * V + 32) >> 6
        for y in range(16):
            row = pic[y0 + y]
            for x in range(16):
                if mode == 0:
                    v = T[x]
                elif mode == 1:
                    v = L[y]
                elif mode == 2:
                    v = dc
                else:
                    v = clip1((a + bb * (x - 7) + cc * (y - 7) + 16) >> 5)
                row[x0 + x] = v

    def _pred_chroma(self, mx, my, mode):
        has_l = self._mb_at_intra(mx - 1, my) is not None
        has_t = self._mb_at_intra(mx, my - 1) is not None
        if (mode == 1 and not has_l) or (mode == 2 and not has_t) or (mode == 3 and not (has_l and has_t and self._mb_at_intra(mx - 1, my - 1) is not None)):
            raise SyntaxError264("chroma prediction mode %d needs neighbours that are not available" % mode)
        x0, y0 = mx * 8, my * 8
        for plane in (self.cur.cb, self.cur.cr):
            T = [plane[y0 - 1][x0 + i] for i in range(8)] if has_t else None
            L = [plane[y0 + i][x0 - 1] for i in range(8)] if has_l else None
            if mode == 0:
                for by in range(2):
                    for bx in range(2):
                        st = sum(T[4 * bx:4 * bx + 4]) if has_t else 0
                        sl = sum(L[4 * by:4 * by + 4]) if has_l else 0
                        if (bx, by) in ((0, 0), (1, 1)):
                            dc = (st + sl + 4) >> 3 if has_t and has_l else (st + 2) >> 2 if has_t else (sl + 2) >> 2 if has_l else 128
                        elif (bx, by) == (1, 0):
                            dc = (st + 2) >> 2 if has_t else (sl + 2) >> 2 if has_l else 128
                        else:
                            dc = (sl + 2) >> 2 if has_l else (st + 2) >> 2 if has_t else 128
                        for y in range(4):
                            for x in range(4):
                                plane[y0 + 4 * by + y][x0 + 4 * bx + x] = dc
                continue
            if mode == 3:
                M = plane[y0 - 1][x0 - 1]
                tt = lambda i: M if i < 0 else T[i]
                ll = lambda i: M if i < 0 else L[i]
                H = sum((k + 1) * (tt(4 + k) - tt(2 - k)) for k in range(4))
                V = sum((k + 1) * (ll(4 + k) - ll(2 - k)) for k in range(4))
                a = 16 * (L[7] + T[7])
                bb, cc = (34 * H + 32) >> 6, (34 * V + 32) >> 6
            for y in range(8):
                for x in range(8):
                    plane[y0 + y][x0 + x] = L[y] if mode == 1 else T[x] if mode == 2 else clip1((a + bb * (x - 3) + cc * (y - 3) + 16) >> 5)

    # ---------------- inter prediction (8.4) ----------------
    def _mv_pred_16x16(self, mx, my):
        """8.4.1.3: median luma motion vector prediction for a 16x16 partition with refIdx 0"""
        A, B = self._mb_at(mx - 1, my), self._mb_at(mx, my - 1)
        C = self._mb_at(mx + 1, my - 1)
        if C is None:
            C = self._mb_at(mx - 1, my - 1)   # D stands in for an unavailable C
            if C is not None:
                self.cover["mvp_D_for_C"] += 1

        def mv_ref(n):
            if n is None or n.intra:
                return (0, 0), -1
            return n.mv, 0
        (mva, ra), (mvb, rb), (mvc, rc) = mv_ref(A), mv_ref(B), mv_ref(C)
        if A is not None and B is None and C is None:
            # only A: refIdxB and refIdxC unavailable -> mvLXA (the partition availability rule of 8.4.1.3.1)
            self.cover["mvp_only_A"] += 1
            return mva
        match = [mv for mv, r in ((mva, ra), (mvb, rb), (mvc, rc)) if r == 0]
        if len(match) == 1:
            self.cover["mvp_single_match"] += 1
            return match[0]
        self.cover["mvp_median"] += 1
        med = lambda a, b, c: a + b + c - min(a, b, c) - max(a, b, c)
        return med(mva[0], mvb[0], mvc[0]), med(mva[1], mvb[1], mvc[1])

    def _p_skip(self, addr):
        mb_w = self.sps["mb_w"]
        mx, my = addr % mb_w, addr // mb_w
        m = self.mbs[addr]
        m.slice = self.slice_no
        A, B = self._mb_at(mx - 1, my), self._mb_at(mx, my - 1)
        if A is None or B is None or (not A.intra and A.mv == (0, 0)) or (not B.intra and B.mv == (0, 0)):
            mv = (0, 0)
            self.cover["skip_mv_zero_rule"] += 1
        else:
            mv = self._mv_pred_16x16(mx, my)
            self.cover["skip_mv_predicted"] += 1
        m.intra, m.i4, m.skip, m.pcm, m.ref, m.mv = False, False, True, False, 0, mv
        m.nz = [0] * 16
        m.nzc = [[0] * 4, [0] * 4]
        m.qp, m.qpc = self.qp, self._qpc(self.qp)
        m.avail = True
        self._inter_pred(m, mx, my)
        self.cover["mb_P_Skip"] += 1

    def _inter_pred(self, m, mx, my):
        ref, pic = self.ref_pic, self.cur
        W, H = ref.w, ref.h
        mvx, mvy = m.mv
        # luma: 8.4.2.2.1, quarter-sample interpolation
        fx, fy = mvx & 3, mvy & 3
        ix, iy = mx * 16 + (mvx >> 2), my * 16 + (mvy >> 2)
        R = ref.y

        def px(x, y):
            return R[clip3(0, H - 1, y)][clip3(0, W - 1, x)]
        if fx == 0 and fy == 0:
            self.cover["mc_luma_integer"] += 1
            for y in range(16):
                row = pic.y[my * 16 + y]
                for x in range(16):
                    row[mx * 16 + x] = px(ix + x, iy + y)
        else:
            self.cover["mc_luma_fractional"] += 1
            tap = lambda a, b, c, d, e, f: a - 5 * b + 20 * c + 20 * d - 5 * e + f

            def b1(x, y):   # intermediate half sample to the right of G = (x, y)
                return tap(px(x - 2, y), px(x - 1, y), px(x, y), px(x + 1, y), px(x + 2, y), px(x + 3, y))

            def h1(x, y):   # below G
                return tap(px(x, y - 2), px(x, y - 1), px(x, y), px(x, y + 1), px(x, y + 2), px(x, y + 3))

            def sample(x, y):
                G = px(x, y)
                if fy == 0:
                    b = clip1((b1(x, y) + 16) >> 5)
                    return b if fx == 2 else (G + b + 1) >> 1 if fx == 1 else (px(x + 1, y) + b + 1) >> 1
                if fx == 0:
                    h = clip1((h1(x, y) + 16) >> 5)
                    return h if fy == 2 else (G + h + 1) >> 1 if fy == 1 else (px(x, y + 1) + h + 1) >> 1
                b = clip1((b1(x, y) + 16) >> 5)
                h = clip1((h1(x, y) + 16) >> 5)
                s = clip1((b1(x, y + 1) + 16) >> 5)
                mm = clip1((h1(x + 1, y) + 16) >> 5)
                j = clip1((tap(b1(x, y - 2), b1(x, y - 1), b1(x, y), b1(x, y + 1), b1(x, y + 2), b1(x, y + 3)) + 512) >> 10)
                if fx == 2 and fy == 2:
                    return j
                if fx == 2:
                    return (b + j + 1) >> 1 if fy == 1 else (s + j + 1) >> 1
                if fy == 2:
                    return (h + j + 1) >> 1 if fx == 1 else (mm + j + 1) >> 1
                # diagonal quarter positions e, g, p, r
                hh = h if fx == 1 else mm
                bb = b if fy == 1 else s
                return (bb + hh + 1) >> 1
            for y in range(16):
                row = pic.y[my * 16 + y]
                for x in range(16):
                    row[mx * 16 + x] = sample(ix + x, iy + y)
        # chroma: 8.4.2.2.2, eighth-sample bilinear
        cfx, cfy = mvx & 7, mvy & 7
        cx, cy = mx * 8 + (mvx >> 3), my * 8 + (mvy >> 3)
        self.cover["mc_chroma_frac_%d_%d" % (cfx, cfy)] += 1
        for src, dst in ((ref.cb, pic.cb), (ref.cr, pic.cr)):
            def cp(x, y):
                return src[clip3(0, H // 2 - 1, y)][clip3(0, W // 2 - 1, x)]
            for y in range(8):
                row = dst[my * 8 + y]
                for x in range(8):
                    a, b = cp(cx + x, cy + y), cp(cx + x + 1, cy + y)
                    c, d = cp(cx + x, cy + y + 1), cp(cx + x + 1, cy + y + 1)
                    row[mx * 8 + x] = ((8 - cfx) * (8 - cfy) * a + cfx * (8 - cfy) * b + (8 - cfx) * cfy * c + cfx * cfy * d + 32) >> 6

    # ---------------- deblocking filter (8.7) ----------------
    def _bs(self, p, q, pblk, qblk, mb_edge):
        """boundary strength between 4x4 luma blocks pblk of macroblock p and qblk of q (index = by * 4 + bx)"""
        if p.intra or q.intra:
            return 4 if mb_edge else 3
        if p.nz[pblk] or q.nz[qblk]:
            return 2
        if p.ref != q.ref or abs(p.mv[0] - q.mv[0]) >= 4 or abs(p.mv[1] - q.mv[1]) >= 4:
            return 1
        return 0

    def _filter_line(self, get, put, bs, qp_av, a_off, b_off, luma):
        ia, ib = clip3(0, 51, qp_av + a_off), clip3(0, 51, qp_av + b_off)
        alpha, beta = ALPHA[ia], BETA[ib]
        p0, p1, q0, q1 = get(-1), get(-2), get(0), get(1)
        if not (abs(p0 - q0) < alpha and abs(p1 - p0) < beta and abs(q1 - q0) < beta):
            return
        self.cover["deblock_%s_bs%d" % ("luma" if luma else "chroma", bs)] += 1
        if bs < 4:
            tc0 = TC0[ia][bs - 1]
            if luma:
                p2, q2 = get(-3), get(2)
                ap, aq = abs(p2 - p0), abs(q2 - q0)
                tc = tc0 + (1 if ap < beta else 0) + (1 if aq < beta else 0)
            else:
                tc = tc0 + 1
            delta = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3)
            put(-1, clip1(p0 + delta))
            put(0, clip1(q0 - delta))
            if luma:
                if ap < beta:
                    put(-2, p1 + clip3(-tc0, tc0, (p2 + ((p0 + q0 + 1) >> 1) - (p1 << 1)) >> 1))
                if aq < beta:
                    put(1, q1 + clip3(-tc0, tc0, (q2 + ((p0 + q0 + 1) >> 1) - (q1 << 1)) >> 1))
            return
        if luma:
            p2, p3, q2, q3 = get(-3), get(-4), get(2), get(3)
            small = abs(p0 - q0) < ((alpha >> 2) + 2)
            if small and abs(p2 - p0) < beta:
                put(-1, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3)
                put(-2, (p2 + p1 + p0 + q0 + 2) >> 2)
                put(-3, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3)
                self.cover["deblock_luma_bs4_strong"] += 1
            else:
                put(-1, (2 * p1 + p0 + q1 + 2) >> 2)
            if small and abs(q2 - q0) < beta:
                put(0, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3)
                put(1, (p0 + q0 + q1 + q2 + 2) >> 2)
                put(2, (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3)
            else:
                put(0, (2 * q1 + q0 + p1 + 2) >> 2)
        else:
            put(-1, (2 * p1 + p0 + q1 + 2) >> 2)
            put(0, (2 * q1 + q0 + p1 + 2) >> 2)

    def _deblock_picture(self):
        sps = self.sps
        mb_w, mb_h = sps["mb_w"], sps["mb_h"]
        pic = self.cur
        for my in range(mb_h):
            for mx in range(mb_w):
                q = self.mbs[my * mb_w + mx]
                idc, a_off, b_off = self.deblock_idc[q.slice]
                if idc == 1:
                    continue
                left = self.mbs[my * mb_w + mx - 1] if mx > 0 else None
                top = self.mbs[(my - 1) * mb_w + mx] if my > 0 else None
                if idc == 2:
                    if left is not None and left.slice != q.slice:
                        left = None
                    if top is not None and top.slice != q.slice:
                        top = None
                # vertical edges, left to right
                for e in range(4):
                    p = left if e == 0 else q
                    if p is None:
                        continue
                    for k in range(16):   # luma rows
                        by = k >> 2
                        bs = self._bs(p, q, by * 4 + (3 if e == 0 else e - 1), by * 4 + e, e == 0)
                        if not bs:
                            continue
                        row = pic.y[my * 16 + k]
                        xe = mx * 16 + 4 * e
                        self._filter_line(lambda i: row[xe + i], lambda i, v: row.__setitem__(xe + i, v), bs,
                                          (p.qp + q.qp + 1) >> 1, a_off, b_off, True)
                    if e & 1:
                        continue
                    for plane in (pic.cb, pic.cr):
                        for k in range(8):
                            by = k >> 1
                            bs = self._bs(p, q, by * 4 + (3 if e == 0 else e - 1), by * 4 + e, e == 0)
                            if not bs:
                                continue
                            row = plane[my * 8 + k]
                            xe = mx * 8 + 2 * e
                            self._filter_line(lambda i: row[xe + i], lambda i, v: row.__setitem__(xe + i, v), bs,
                                              (p.qpc + q.qpc + 1) >> 1, a_off, b_off, False)
                # horizontal edges, top to bottom
                for e in range(4):
                    p = top if e == 0 else q
                    if p is None:
                        continue
                    for k in range(16):   # luma columns
                        bx = k >> 2
                        bs = self._bs(p, q, (3 if e == 0 else e - 1) * 4 + bx, e * 4 + bx, e == 0)
                        if not bs:
                            continue
                        x, ye = mx * 16 + k, my * 16 + 4 * e
                        self._filter_line(lambda i: pic.y[ye + i][x], lambda i, v: pic.y[ye + i].__setitem__(x, v), bs,
                                          (p.qp + q.qp + 1) >> 1, a_off, b_off, True)
                    if e & 1:
                        continue
                    for plane in (pic.cb, pic.cr):
                        for k in range(8):
                            bx = k >> 1
                            bs = self._bs(p, q, (3 if e == 0 else e - 1) * 4 + bx, e * 4 + bx, e == 0)
                            if not bs:
                                continue
                            x, ye = mx * 8 + k, my * 8 + 2 * e
                            self._filter_line(lambda i: plane[ye + i][x], lambda i, v: plane[ye + i].__setitem__(x, v), bs,
                                              (p.qpc + q.qpc + 1) >> 1, a_off, b_off, False)
