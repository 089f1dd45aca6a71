"""The schedule policy of a session (cedarx_h264_encoder_amd/csrc/enc_plan.h) through cedar_enc_plan(): pure
functions of geometry, CU count, probe results and the CEDAR_* switches, so every threshold is held here on a box
without a GPU -- pipelining on / off, deblock wave pairs, cohorts, planes, the widening of the deblock split
(round 2's 4K deadlock lived in that code), stripes and how many are held back. The reference has no counterpart:
its one silicon pipeline runs one frame at a time (kernel/cedar.c:1176-1180)."""
import ctypes

import pytest

import cedarx_h264_encoder_amd as ce


class Query(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("mb_w", "mb_h", "slice_count", "slice_first", "slice_num", "n_streams", "cus",
                                            "db_streams_usable", "holds_pipe_token")]


class Result(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("error", "depth", "wf_split", "wf_split_intra", "db_pairs", "split_planes",
                                            "want_row_pipe", "db_streams_wanted", "row_pipe", "db_streams", "inter_bands",
                                            "defer_default", "cohorts", "deblock_rows", "intra_rows", "db_wgs_per_xcd", "db_lds_kb",
                                            "idr_beside_ok")]


ENV = ("CEDAR_ROW_PIPE", "CEDAR_PIPE_DEPTH", "CEDAR_WF_SPLIT", "CEDAR_WF_SPLIT_INTRA", "CEDAR_DB_PAIRS", "CEDAR_DEBLOCK_STREAMS",
       "CEDAR_DEBLOCK_PLANES", "CEDAR_INTER_BANDS", "CEDAR_DEFER", "CEDAR_COHORTS", "CEDAR_DEBLOCK_ROWS", "CEDAR_INTRA_ROWS")


@pytest.fixture
def plan(monkeypatch):
    for k in ENV:
        monkeypatch.delenv(k, raising=False)
    L = ce.load_library()
    L.cedar_enc_plan.argtypes = [ctypes.POINTER(Query), ctypes.POINTER(Result)]

    def run(w=1920, h=1080, streams=1, slices=1, first=0, num=0, cus=256, usable=3, token=1, **env):
        for k, v in env.items():
            monkeypatch.setenv("CEDAR_" + k.upper(), str(v))
        q = Query((w + 15) // 16, (h + 15) // 16, slices, first, num, streams, cus, usable, token)
        r = Result()
        rc = L.cedar_enc_plan(ctypes.byref(q), ctypes.byref(r))
        for k in env:
            monkeypatch.delenv("CEDAR_" + k.upper())
        assert rc == 0
        return r
    return run


def test_few_units_deblock_with_a_wave_per_macroblock_row(plan):
    """k_deblock_rows: a workgroup per band of 4 rows and plane, as long as two pictures' worth of them leave the analysis
    most of the chip; two deblock streams, nothing held back. Many units keep the wave pairs of k_deblock2_planes."""
    r = plan()
    assert r.deblock_rows == 1 and (r.want_row_pipe, r.db_streams_wanted) == (1, 2)
    assert (r.row_pipe, r.db_streams, r.inter_bands, r.defer_default, r.cohorts) == (1, 2, 5, 0, 1)   # (5 stripes since round 4's back end)
    assert plan(w=3840, h=2160).inter_bands == 5 and plan(w=854, h=480).inter_bands == 3 and plan(inter_bands=4).inter_bands == 4
    assert plan(w=3840, h=2160, slices=4).inter_bands == 3 and plan(streams=2).inter_bands == 3   # (several units on this kernel keep 3)
    assert plan(usable=1).db_streams == 1 and plan(usable=0).row_pipe == 0
    for kw, rows, pipe in ((dict(w=854, h=480), 1, 1), (dict(w=3840, h=2160), 1, 1), (dict(w=3840, h=2160, slices=4), 1, 1),   # 2 pictures: 32 / 136 / 144 workgroups
                           (dict(streams=2), 1, 1), (dict(streams=3), 0, 1),                                                      # 136 / 204 (3/4 x 256 = 192): wave pairs
                           (dict(streams=4), 1, 0), (dict(streams=5), 1, 0), (dict(streams=8), 1, 0), (dict(streams=23), 1, 0),   # 4 .. 23 units: rows, frames back to back
                           (dict(streams=24), 0, 0), (dict(streams=32), 0, 0),
                           (dict(w=3840, h=2160, slices=4, streams=2), 1, 0), (dict(w=320, h=240, streams=8), 1, 1), (dict(w=320, h=240, streams=9), 1, 0)):
        r = plan(**kw)
        assert (r.deblock_rows, r.want_row_pipe) == (rows, pipe), kw
    assert [plan(streams=n).depth for n in (6, 8, 9, 16, 23, 24)] == [10, 8, 6, 6, 6, 4]   # (<= 8 units: the 512 MB ring; 9 .. 23 on this kernel: 6)
    assert plan(deblock_rows=0).deblock_rows == 0 and plan(streams=32, deblock_rows=2).deblock_rows == 1
    assert plan(streams=3, deblock_rows=1).deblock_rows == 0          # (1 = the default rule)
    assert plan(cus=64).deblock_rows == 0 and plan(cus=96).deblock_rows == 1   # 2 x 34 workgroups on 3/4 of the CUs
    assert plan(cus=96).want_row_pipe == 1 and plan(streams=8, deblock_rows=0).want_row_pipe == 1
    assert plan(deblock_streams=3).db_streams_wanted == 3


def test_one_1080p_stream_is_pipelined_with_held_back_stripes(plan, monkeypatch):
    monkeypatch.setenv("CEDAR_DEBLOCK_ROWS", "0")   # the wave-pair kernel's policy
    r = plan()
    assert (r.depth, r.wf_split, r.db_pairs, r.split_planes) == (16, 8, 8, 1)
    assert (r.want_row_pipe, r.db_streams_wanted) == (1, 3)
    assert (r.row_pipe, r.db_streams, r.inter_bands, r.defer_default, r.cohorts) == (1, 3, 8, 3, 1)
    assert r.wf_split_intra == 9            # 17 bands of 4 rows, two triples per workgroup
    # only two deblock streams really overlap: no third one, nothing held back, 5 stripes as before
    r = plan(usable=2)
    assert (r.row_pipe, r.db_streams, r.inter_bands, r.defer_default) == (1, 2, 5, 0)
    # no stream overlaps, or another session of the process holds the device's pipelining: frames back to back
    for kw in (dict(usable=0), dict(token=0)):
        r = plan(**kw)
        assert (r.row_pipe, r.db_streams, r.inter_bands, r.defer_default) == (0, 1, 1, 0)
    # switches
    assert plan(defer=0).db_streams_wanted == 2 and plan(defer=0).defer_default == 0
    assert plan(inter_bands=6).defer_default == 2 and plan(inter_bands=5).defer_default == 0
    assert plan(row_pipe=0).row_pipe == 0 and plan(row_pipe=0).depth == 4
    assert plan(pipe_depth=7).depth == 7


def test_small_pictures_take_three_stripes_and_hold_nothing_back(plan, monkeypatch):
    monkeypatch.setenv("CEDAR_DEBLOCK_ROWS", "0")
    r = plan(w=854, h=480)
    assert (r.row_pipe, r.db_streams, r.inter_bands, r.defer_default) == (1, 3, 3, 0)
    r = plan(w=320, h=240)
    assert (r.wf_split, r.wf_split_intra) == (2, 2)   # 4 bands: a split never leaves a workgroup fewer than two
    r = plan(w=3840, h=2160)
    assert (r.row_pipe, r.inter_bands, r.defer_default, r.depth) == (1, 8, 3, 16)   # 33 MB of pinned bytestream a slot: 16 of them just fit 512 MB


def test_units_decide_the_deblock_form_the_split_and_the_pipelining(plan, monkeypatch):
    monkeypatch.setenv("CEDAR_DEBLOCK_ROWS", "0")
    want = {  # streams: (wf_split, db_pairs, want_row_pipe, split_planes, cohorts)
        1: (8, 8, 1, 1, 1), 2: (8, 8, 1, 1, 1), 3: (4, 8, 1, 1, 1), 5: (4, 8, 1, 1, 1), 6: (2, 8, 1, 1, 1), 8: (2, 8, 1, 1, 1),
        9: (9, 2, 0, 1, 1), 16: (9, 2, 0, 1, 1), 31: (9, 2, 0, 1, 1), 32: (9, 2, 0, 0, 2), 64: (9, 2, 0, 0, 2), 200: (9, 2, 0, 0, 2),
    }
    for s, w in want.items():
        r = plan(streams=s, usable=2)
        assert (r.wf_split, r.db_pairs, r.want_row_pipe, r.split_planes, r.cohorts) == w, s
        slot = 8160 * 1024 + 4096   # pinned bytestream per stream and ring slot; the ring stays under 512 MB
        assert r.depth == (max(4, min(16, (512 << 20) // (slot * s))) if s <= 8 else 4), s
        assert r.db_streams_wanted == (3 if s == 1 else 2)
    # pipelining must leave half of the CUs to the analysis the resident deblock workgroups wait for
    assert plan(streams=4, cus=256).want_row_pipe == 1 and plan(streams=4, cus=120).want_row_pipe == 0
    assert plan(streams=32, cohorts=3).cohorts == 3 and plan(streams=2, cohorts=4).cohorts == 1   # (pipelined: one)
    assert plan(streams=32, db_pairs=8).cohorts == 1 and plan(streams=32, deblock_planes=1).split_planes == 1


def test_a_band_needs_a_wave_pair_where_rows_are_long(plan):
    """A band runs at most 16 (+3) macroblocks ahead of the band below it (the LDS ring); a workgroup with more bands
    than wave pairs starts band b + pairs when band b is through: on a row longer than pairs x 16 macroblocks that is a
    cycle -- 3840 wide, 9 bands on 8 pairs deadlocked in round 2. The split widens until every band has a pair."""
    r = plan(w=3840, h=592, wf_split=1)                # 240 x 37 macroblocks, 10 bands, asked to run as ONE workgroup
    assert r.error == 0 and (10 + r.wf_split - 1) // r.wf_split <= r.db_pairs
    r = plan(w=3840, h=2160, streams=16)               # 34 bands, small form: 17 workgroups would do, 16 is the most
    assert r.error == 0 and r.db_pairs == 8 and (34 + r.wf_split - 1) // r.wf_split <= 8
    r = plan(w=1920, h=1080, streams=16)               # 120 + 3 <= 2 x 16? no: every band its own pair
    assert (r.db_pairs, r.wf_split) == (2, 9) and (17 + 8) // 9 <= 2
    r = plan(w=256, h=4352, wf_split=1)                # short rows: 16 + 3 <= 8 x 16, one workgroup may serialise its 68 bands
    assert r.error == 0 and r.wf_split == 1
    assert plan(w=3840, h=16 * 4 * 8 * 16 + 64).error == 1   # 513 bands: more than 16 workgroups x 8 pairs can give a pair each


def test_slices_and_slice_shards(plan, monkeypatch):
    monkeypatch.setenv("CEDAR_DEBLOCK_ROWS", "0")
    r = plan(w=3840, h=2160, slices=4)                 # configs[3] on one GPU: 4 units of 34 / 34 / 34 / 33 rows
    assert (r.wf_split, r.db_pairs, r.want_row_pipe, r.db_streams_wanted, r.defer_default) == (4, 8, 1, 2, 0)
    assert r.wf_split_intra == 5                       # 9 bands, two triples per workgroup
    r = plan(w=3840, h=2160, slices=4, first=3, num=1)  # one slice of four on its own GPU: one unit of 33 rows
    assert (r.wf_split, r.db_pairs, r.want_row_pipe, r.db_streams_wanted) == (4, 8, 1, 3)
    assert (r.inter_bands, r.defer_default) == (8, 3)   # 240 x 33 macroblocks: held-back stripes as for any one unit
    L = ce.load_library()
    bad = Query(120, 68, 4, 3, 2, 1, 256, 2, 1)
    assert L.cedar_enc_plan(ctypes.byref(bad), ctypes.byref(Result())) < 0


def test_resident_deblock_workgroups_always_leave_every_xcd_room(plan):
    """Round 3's hang of 4 to 8 pipelined 1080p streams on k_deblock_rows, and round 4's one 3840x2160 session in ten that
    ended in -EIO: a pipelined deblock kernel's workgroups are RESIDENT and poll for their picture's analysis, every band
    claims enough LDS to have its CU to itself (81 KB), the hardware deals a launch's workgroups round-robin over the 8
    XCDs and each XCD places its share in order on its own 32 CUs -- so the analysis (and an IDR picture's intra
    wavefront) stalls as soon as ONE XCD has no CU left, however empty the rest of the chip is: three pictures of 72
    workgroups and 72 intra workgroups were 288 on 256. The guard is counted per XCD: the pictures that can be resident
    at a time -- one per deblock stream, a third only where idr_beside_ok says so -- leave two CUs of every XCD alone."""
    seen_pipelined = seen_refused = 0
    for w, h in ((320, 240), (854, 480), (1280, 720), (1920, 1080), (2560, 1440), (3840, 2160), (4096, 2304), (1920, 4320)):
        for streams in (1, 2, 3, 4, 5, 6, 8):
            for slices in (1, 2, 4):
                for cus in (256, 128, 64):
                    r = plan(w=w, h=h, streams=streams, slices=slices, cus=cus)
                    if not (r.deblock_rows and r.row_pipe):
                        seen_refused += 1
                        continue
                    seen_pipelined += 1
                    xcd = cus // 8
                    assert r.db_lds_kb == 81                      # a CU per band, room for one intra workgroup (<= 79 KB) beside it
                    assert r.db_streams <= 2
                    assert r.db_streams * r.db_wgs_per_xcd <= xcd - 2, (w, h, streams, slices, cus)
                    if r.idr_beside_ok:
                        assert 3 * r.db_wgs_per_xcd <= xcd - 2, (w, h, streams, slices, cus)
    assert seen_pipelined > 20 and seen_refused > 20
    # the sessions of the bench line: one 1080p stream and the 4K picture may have an IDR picture's deblocking beside its intra
    # wavefront only where three pictures fit
    assert plan().idr_beside_ok == 1 and plan().db_wgs_per_xcd == 5               # 34 workgroups a picture
    assert plan(w=3840, h=2160).idr_beside_ok == 1 and plan(w=3840, h=2160).db_wgs_per_xcd == 9   # 68 a picture: 3 x 9 <= 30
    r = plan(w=2560, h=1440, streams=2)                                            # 92 a picture: 2 x 12 <= 30 < 3 x 12
    assert (r.deblock_rows, r.row_pipe, r.db_wgs_per_xcd, r.idr_beside_ok) == (1, 1, 12, 0)
    assert plan(streams=3).row_pipe == 1 and plan(streams=3).deblock_rows == 0     # 3 x 1080p: 102 a picture, 2 x 102 > 3/4 of the CUs -> the wave pairs
    assert plan(streams=4).row_pipe == 0 and plan(streams=4).deblock_rows == 1     # 4 x 1080p: 2 x 17 > 30 on an XCD -> this kernel, frames back to back
    # back to back nothing waits: no claim
    assert plan(streams=8).deblock_rows == 1 and plan(streams=8).row_pipe == 0 and plan(streams=8).db_lds_kb == 16
