"""Parity tests proper: the HIP path, called through the C-ABI, against the CPU
oracle on the same seeded inputs. Integer pipeline => the bar is BIT-EXACT:
identical Annex-B bytes, identical per-macroblock decisions/levels and
identical deblocked reconstruction. All tests here need an MI355X."""
import errno
import os
import sys

import numpy as np
import pytest

from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ce():
    import cedarx_h264_encoder_amd as m
    return m


CASES = [
    # w, h, qp, keyint, frames, pattern, slices
    (64, 48, 30, 1, 3, "B", 1),      # I only
    (320, 240, 30, 1, 3, "B", 1),    # BASELINE configs[0] shape
    (64, 48, 30, 25, 5, "B", 1),
    (64, 48, 24, 25, 4, "C", 1),     # noise: every block coded, long levels
    (64, 48, 30, 25, 4, "A", 1),     # static: P pictures are one skip run
    (70, 50, 24, 25, 4, "B", 1),     # ragged size: padding + SPS cropping
    (320, 240, 30, 25, 6, "B", 1),
    (320, 240, 24, 25, 4, "B", 4),   # 4 slices, deblocking idc 2 (configs[3] structure)
    (48, 80, 24, 4, 6, "B", 5),
    (64, 48, 5, 3, 4, "C", 1),       # very low QP: escape-coded levels
    (64, 48, 47, 3, 4, "B", 1),      # highest QP the reference accepts
    (16, 16, 24, 2, 4, "C", 1),      # single macroblock
    (16, 350, 30, 1, 4, "A", 3),     # one macroblock wide, tall: no neighbour C anywhere (k_cavlc2 read record -1 of the second row)
    (16, 96, 24, 3, 6, "B", 2),
    (432, 32, 24, 3, 4, "B", 1),
    (64, 80, 24, 3, 4, "B", 1),      # 5 macroblock rows: idle row groups in the last band
    (96, 112, 30, 25, 4, "C", 1),    # 7 rows
    (854, 480, 30, 25, 3, "B", 1),   # BASELINE configs[1] shape (README's 854x480)
]


_ORACLE_STREAMS = {}


def _oracle_stream(w, h, qp, keyint, n, pattern="B"):
    """(frames, the scalar twin's access units) of a plain session; kept for the tests that share it (the twin takes
    0.7 s per 1080p frame): a request for fewer frames of a session already made is a prefix of it"""
    key = (w, h, qp, keyint, pattern)
    have = _ORACLE_STREAMS.get(key)
    if have is None or len(have[0]) < n:
        frames = [ol.gen_frame(pattern, w, h, f) for f in range(n)]
        oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=keyint))
        while len(_ORACLE_STREAMS) >= 3:   # (30 frames of 1080p are 90 MB)
            del _ORACLE_STREAMS[next(iter(_ORACLE_STREAMS))]
        have = _ORACLE_STREAMS[key] = (frames, [oe.encode(y, uv) for y, uv in frames])
    else:
        _ORACLE_STREAMS[key] = _ORACLE_STREAMS.pop(key)   # most recently used last
    return have[0][:n], have[1][:n]


def _compare_state(oe, ge):
    om = oe.mb_array()
    gm = ge.debug_read("mbinfo")
    for fld in ("type", "i16_mode", "chroma_mode", "cbp", "mvx", "mvy"):
        assert np.array_equal(om[fld], gm[fld]), fld
    assert np.array_equal(om["i4_modes"], ge.debug_read("i4modes"))
    assert np.array_equal(om["nnz"], ge.debug_read("nnz"))
    gc = ge.debug_read("coef")
    n = len(om)
    assert np.array_equal(om["luma"].reshape(n, 256), gc[:, 0:256])
    assert np.array_equal(om["dc16"], gc[:, 256:272])
    assert np.array_equal(om["cdc"].reshape(n, 8), gc[:, 272:280])
    assert np.array_equal(om["cac"].reshape(n, 8, 16)[:, :, 1:], gc[:, 280:408].reshape(n, 8, 16)[:, :, 1:])
    ry, ruv = oe.recon()
    assert np.array_equal(ry, ge.debug_read("recon_y"))
    assert np.array_equal(ruv, ge.debug_read("recon_uv"))


@pytest.mark.parametrize("w,h,qp,keyint,frames,pattern,slices", CASES + [(64, 48, 5, 3, 4, "C", -1)])
def test_bitstream_and_state_equal_oracle(ce, w, h, qp, keyint, frames, pattern, slices):
    kw = {}
    if slices < 0:   # I_PCM switched off: noise at a very low QP keeps its escape-coded levels (and passes 3200 bits)
        slices, kw = 1, dict(pcm=-1)
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=keyint, slice_count=slices, **kw))
    with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint, slice_count=slices, **kw) as ge:
        for f in range(frames):
            y, uv = ol.gen_frame(pattern, w, h, f)
            ref = oe.encode(y, uv)
            got = ge.encode_frame(y, uv)
            _compare_state(oe, ge)
            assert got == ref, "frame %d: bytestream differs" % f


@pytest.mark.parametrize("w,h,qp,keyint,patterns,frames,kw", [
    (96, 64, 1, 3, "C", 4, {}), (96, 64, 16, 3, "C", 4, {}), (96, 64, 18, 3, "C", 5, {}), (96, 64, 20, 25, "C", 4, {}),
    (96, 80, 12, 4, "CB", 6, dict(slice_count=2)), (64, 48, 8, 25, "C", 3, dict(subpel=2)), (16, 48, 6, 2, "C", 3, {}),
    (320, 240, 17, 5, "C", 7, dict(slice_count=3)), (640, 368, 18, 25, "CCB", 4, {}),
    (1920, 1080, 1, 25, "C", 2, {}), (96, 64, 10, 3, "C", 6, dict(n_streams=3)), (96, 64, 14, 4, "C", 9, dict(gop_parallel=2)),
    (96, 64, 5, 3, "C", 3, dict(pcm=-1)), (96, 64, 5, 3, "C", 3, dict(intra_refresh=3)), (96, 64, 22, 3, "C", 3, {})])
def test_macroblocks_that_could_pass_3200_bits_are_coded_i_pcm(ce, w, h, qp, keyint, patterns, frames, kw):
    """cedar_enc_config_ext.pcm (default on): a macroblock whose CAVLC bits could pass the 3200 of H.264 A.3.1 is coded
    I_PCM. k_intra_decide counts on the open loop's levels (a tenth of room), k_inter on the final ones; the intra
    wavefront copies the source, both CAVLC kernels write mb_type only, k_scan places the samples at the next byte
    boundary, k_pack copies them from the source planes, the deblock kernel leaves the macroblock's edges alone.
    Records, counts, levels, reconstruction and bytes equal the scalar twin's at 1080p, in slices, lock-step
    streams, GOP lanes and with frames in flight; the independent decoder rebuilds the pictures and finds no
    macroblock above the limit."""
    from tests.indep import h264_refdec as rd
    S, G = kw.get("n_streams", 1), kw.get("gop_parallel", 0)
    okw = {k: v for k, v in kw.items() if k not in ("n_streams", "gop_parallel")}
    src = [[ol.gen_frame(patterns[f % len(patterns)], w, h, f, s << 8) for s in range(S)] for f in range(frames)]
    oes = [ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=keyint, **okw)) for _ in range(S)]
    want = [[oes[s].encode(*src[f][s]) for f in range(frames)] for s in range(S)]
    if G:
        with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint, **kw) as ge:
            got = _encode_gop_parallel(ge, [src[f][0] for f in range(frames)], streams=1)
        assert got == want[0]
        return
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=keyint, **okw))
    dec = rd.Decoder() if w <= 320 else None
    n_pcm = 0
    with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint, **kw) as ge:
        for f in range(frames):
            for s in range(S):
                ge.set_input(src[f][s][0], src[f][s][1], stream=s)
            ge.encode()
            oe.encode(*src[f][0])
            _compare_state(oe, ge)
            n_pcm += int((oe.mb_array()["type"] == ol.MB_IPCM).sum())
            for s in range(S):
                assert ge.output(s) == want[s][f], "frame %d stream %d: bytestream differs" % (f, s)
            if dec:
                pic, = dec.decode(ge.output(0))
                assert pic.luma_bytes() == ge.debug_read("recon_y").tobytes()
                assert pic.nv12_chroma_bytes() == ge.debug_read("recon_uv").tobytes()
        takes = "pcm" not in kw and "intra_refresh" not in kw and qp <= 20
        assert (n_pcm > 0) == takes, n_pcm
        if takes:
            assert ge.stats()["oversize_frames"] == 0
            if dec:
                assert dec.max_mb_bits <= 3200 and dec.cover["mb_I_PCM"] == n_pcm and not dec.level_violations()
    if S == 1:   # frames in flight
        with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint, **kw) as ge:
            assert _encode_pipelined(ge, [src[f][0] for f in range(frames)]) == want[0]


@pytest.mark.parametrize("w,h,qp,keyint,before,after,cuts,kw", [
    (320, 240, 30, 25, "B", "A", (5,), {}), (320, 240, 30, 25, "A", "B", (1, 9), {}), (640, 368, 28, 6, "B", "A", (7,), dict(subpel=2)),
    (320, 240, 30, 25, "A", "B", (3,), dict(slice_count=2)), (1920, 1080, 24, 25, "B", "A", (4,), {}), (320, 240, 28, 25, "B", "C", (2,), {})])
def test_scene_cuts_become_idr_pictures_like_the_oracles(ce, w, h, qp, keyint, before, after, cuts, kw):
    """cedar_enc_config_ext.scene_cut: k_inter counts the macroblocks of a P picture that their own mean predicts better
    than the best vector, k_copy_out reports the count with the frame, and collect codes the picture again as an IDR
    picture that starts a new GOP -- the frames in flight behind it after it (the replay of a failed frame). Bytes equal
    the scalar twin's, synchronous and with the ring full; cuts right after an IDR picture and across a GOP boundary."""
    n = max(cuts) + (ce.load_library().cedar_enc_pipeline_depth() + 4 if w <= 640 else 5)   # (the ring really full at the small sizes)
    def pat(f):
        k = sum(1 for c in cuts if f >= c)
        return (before, after)[k & 1]
    frames = [ol.gen_frame(pat(f), w, h, f) for f in range(n)]
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=keyint, scene_cut=50, **kw))
    want, n_cut = [], 0
    for f, (y, uv) in enumerate(frames):
        want.append(oe.encode(y, uv))
        n_cut += int(ol.lib().h264o_last_frame_type(oe.h) == 1 and f > 0 and (after == "C" or f in cuts))
    assert n_cut >= len(cuts)
    with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint, scene_cut=50, **kw) as enc:
        got = [enc.encode_frame(y, uv) for y, uv in frames[:max(cuts) + 3]]
        assert got == want[:len(got)]
        assert enc.stats()["scene_cuts"] >= sum(1 for c in cuts if c < len(got))
    with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint, scene_cut=50, **kw) as enc:
        assert _encode_pipelined(enc, frames) == want
        assert enc.stats()["scene_cuts"] >= len(cuts)
    for bad in (dict(scene_cut=101), dict(scene_cut=50, n_streams=2), dict(scene_cut=50, gop_parallel=2),
                dict(scene_cut=50, slice_count=2, slice_first=0, slice_num=1)):
        with pytest.raises(ce.EncoderError):
            ce.CedarEncoder(width=w, height=h, **bad)


def test_randomised_geometries_qps_slices(ce):
    """Seeded sweep over odd shapes: widths/heights that are not multiples of 16, row counts
    that are not multiples of 4 (idle row groups in the wavefront kernels), several slices,
    workgroup splits of the wavefront, tiny search ranges, extreme QPs."""
    rng = np.random.RandomState(20261003)
    for case in range(24):
        w = int(rng.randint(1, 20)) * 16 - int(rng.choice([0, 0, 2, 6, 10, 14]))
        h = int(rng.randint(1, 14)) * 16 - int(rng.choice([0, 0, 2, 4, 8, 12]))
        w, h = max(w, 16), max(h, 16)
        mb_h = (h + 15) // 16
        qp = int(rng.choice([1, 8, 17, 24, 30, 36, 41, 47]))
        keyint = int(rng.choice([1, 2, 3, 25]))
        slices = int(rng.randint(1, min(4, mb_h) + 1))
        search = int(rng.choice([0, 0, 0, 4, -1]))
        pat = str(rng.choice(["A", "B", "C"]))
        oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=keyint, slice_count=slices,
                                                search_range=16 if search == 0 else max(search, 0)))
        with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint, slice_count=slices,
                             search_range=search) as ge:
            for f in range(3):
                y, uv = ol.gen_frame(pat, w, h, f)
                ref = oe.encode(y, uv)
                got = ge.encode_frame(y, uv)
                assert got == ref, "case %d (%dx%d qp%d ki%d sl%d sr%d %s) frame %d" % (
                    case, w, h, qp, keyint, slices, search, pat, f)
                ry, ruv = oe.recon()
                assert np.array_equal(ry, ge.debug_read("recon_y"))
                assert np.array_equal(ruv, ge.debug_read("recon_uv"))


def test_1080p_gop_bit_exact_and_decodable(ce):
    """BASELINE configs[2] at full size (1920x1080 I+P QP 24): a short run against
    the oracle, then a longer run checked by the size-independent property
    decode(stream) == in-loop reconstruction."""
    w, h = 1920, 1080
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=24, keyint=25))
    dec = ol.OracleDecoder()
    with ce.CedarEncoder(width=w, height=h, qp=24, keyframe_interval=25) as ge:
        for f in range(8):
            y, uv = ce.gen_pattern("B", w, h, f)
            got = ge.encode_frame(y, uv)
            if f < 3:
                assert got == oe.encode(y, uv), "frame %d" % f
            assert dec.decode(got) == 1
            dy, duv = dec.picture()
            assert np.array_equal(dy, ge.debug_read("recon_y"))
            assert np.array_equal(duv, ge.debug_read("recon_uv"))
        assert dec.crop_size() == (w, h)


def test_1080p_crosses_a_gop_boundary_of_25_frames(ce):
    """BASELINE configs[2] as it is benchmarked (QP 24, keyint 25, the full ring in flight): the 27
    access units around the first GOP boundary against the scalar oracle, every one decodable."""
    w, h, n = 1920, 1080, 27
    frames, want = _oracle_stream(w, h, 24, 25, 30)
    frames, want = frames[:n], want[:n]
    dec = ol.OracleDecoder()
    with ce.CedarEncoder(width=w, height=h, qp=24, keyframe_interval=25) as enc:
        got = _encode_pipelined(enc, frames)
        ry, ruv = enc.debug_read("recon_y"), enc.debug_read("recon_uv")
    for f in range(n):
        assert got[f] == want[f], f
        assert dec.decode(got[f]) == 1
    kinds = [got[25][i + 4] & 0x1f for i in range(len(got[25]) - 4) if got[25][i:i + 4] == b"\0\0\0\1"]
    assert kinds == [5]                       # the second IDR picture: no parameter sets again (cedar.c:1058-1061)
    dy, duv = dec.picture()
    assert np.array_equal(dy, ry) and np.array_equal(duv, ruv)


def test_4k_four_slices_decodable_and_matches_oracle_first_frames(ce):
    """BASELINE configs[3] structure on one device: 3840x2160 in 4 slices (34/34/34/33 rows)."""
    w, h = 3840, 2160
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=24, keyint=25, slice_count=4))
    dec = ol.OracleDecoder()
    with ce.CedarEncoder(width=w, height=h, qp=24, keyframe_interval=25, slice_count=4) as ge:
        for f in range(3):
            y, uv = ce.gen_pattern("B", w, h, f)
            got = ge.encode_frame(y, uv)
            if f < 2:
                assert got == oe.encode(y, uv), "frame %d" % f
            assert dec.decode(got) == 1
            dy, duv = dec.picture()
            assert np.array_equal(dy, ge.debug_read("recon_y"))
            assert np.array_equal(duv, ge.debug_read("recon_uv"))


def test_4k_single_slice_row_count_not_multiple_of_four(ce):
    """3840x2160 as ONE slice: 135 macroblock rows = 33 bands of 4 + 3, so the last
    band of both wavefront kernels has an idle row group (regression: its prefetch
    once indexed row -1). Checked by decode(stream) == reconstruction."""
    w, h = 3840, 2160
    dec = ol.OracleDecoder()
    with ce.CedarEncoder(width=w, height=h, qp=24, keyframe_interval=25) as ge:
        for f in range(4):
            y, uv = ce.gen_pattern("B", w, h, f)
            got = ge.encode_frame(y, uv)
            assert dec.decode(got) == 1
            dy, duv = dec.picture()
            assert np.array_equal(dy, ge.debug_read("recon_y"))
            assert np.array_equal(duv, ge.debug_read("recon_uv"))


@pytest.mark.parametrize("w,h,keyint,defer,dbs", [(1920, 1080, 25, None, None), (1920, 1080, 25, "2", "3"), (1920, 1080, 25, "0", None),
                                                  (1920, 1080, 4, "2", "2"), (854, 480, 25, "3", "3"), (320, 240, 5, "2", "2")])
def test_held_back_stripes_give_the_same_bytes(ce, monkeypatch, w, h, keyint, defer, dbs):
    """The last stripes of a pipelined P picture's analysis and its entropy launches are enqueued between the first
    stripes of the NEXT picture, so that they do not sit in front of them on the in-order analysis stream (DESIGN.md
    5.2): the default for one 1080p stream (first case), CEDAR_DEFER / CEDAR_DEBLOCK_STREAMS / CEDAR_INTER_BANDS for
    the rest. Same bytes, with frames kept in flight and collected one by one, across IDR pictures, with the tail
    flushed by collect when no further submit comes."""
    if defer is not None:
        monkeypatch.setenv("CEDAR_DEFER", defer)
        monkeypatch.setenv("CEDAR_INTER_BANDS", "5" if w >= 1920 else "3")
    if dbs is not None:
        monkeypatch.setenv("CEDAR_DEBLOCK_STREAMS", dbs)
    n = 30 if keyint == 25 else 14
    qp = 24 if w >= 1920 else 27
    frames, want = _oracle_stream(w, h, qp, keyint, n)
    with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint) as enc:
        got = _encode_pipelined(enc, frames)
        assert got == want
    # one at a time: every frame's held-back part is flushed by its own collect
    with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint) as enc:
        assert [enc.encode_frame(y, uv) for y, uv in frames[:8]] == want[:8]


def test_a_caller_may_synchronise_the_device_between_submits(ce):
    """The caller only ever blocks in ENCODE (cedar.c:1178-1180); between two calls it may do anything, a device-wide
    synchronisation included. A session that holds the tail of a P picture back (kernels on the device then wait for
    work only the next submit enqueues) has a thread that enqueues the tail ~100 us after the submit when no further
    submit has come: torch.cuda.synchronize() between submits returns in milliseconds, not after the kernels' 3 s
    bound, and the bytes are the oracle's."""
    import time
    import torch
    w, h, n = 1920, 1080, 12
    frames, want = _oracle_stream(w, h, 24, 25, n)
    got, worst = [], 0.0
    with ce.CedarEncoder(width=w, height=h, qp=24, keyframe_interval=25) as enc:
        q = 0
        for f, (y, uv) in enumerate(frames):
            if q == 4:
                enc.collect(); got.append(enc.output()); q -= 1
            enc.set_input(y, uv)
            enc.submit(); q += 1
            if f % 3 == 1:
                time.sleep(0.002)         # the caller pauses ...
            t0 = time.perf_counter()
            torch.cuda.synchronize()      # ... and waits for the whole device
            worst = max(worst, time.perf_counter() - t0)
        while q:
            enc.collect(); got.append(enc.output()); q -= 1
    assert got == want
    assert worst < 0.5, "a device-wide synchronisation between submits took %.3f s" % worst


@pytest.mark.parametrize("pairs", ["8", "2"])
def test_wide_picture_with_more_bands_than_wave_pairs(ce, monkeypatch, pairs):
    """3840x592 (240 x 37 macroblocks, 10 deblock bands) asked to run as ONE deblock workgroup: a band
    may only run 16 (+3) macroblocks ahead of the band below (the LDS hand-off ring) while band b + pairs
    waits for band b's wave pair, so on a row this long ten bands on eight (or two) pairs would wait for
    each other forever (-EIO after the spin bound). The session widens the split until every band has a pair."""
    monkeypatch.setenv("CEDAR_WF_SPLIT", "1")
    monkeypatch.setenv("CEDAR_DB_PAIRS", pairs)
    w, h = 3840, 592
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=30, keyint=25))
    with ce.CedarEncoder(width=w, height=h, qp=30, keyframe_interval=25) as ge:
        for f in range(3):
            y, uv = ol.gen_frame("B", w, h, f)
            assert ge.encode_frame(y, uv) == oe.encode(y, uv), f


def test_slices_on_separate_sessions_concatenate_to_the_whole(ce):
    """configs[3] sharding: each 'GPU' encodes one slice with its own session; the
    host concatenates the NALs. Must equal the single-session 4-slice stream."""
    w, h, n = 320, 240, 4
    whole = ce.CedarEncoder(width=w, height=h, qp=24, keyframe_interval=25, slice_count=n)
    parts = [ce.CedarEncoder(width=w, height=h, qp=24, keyframe_interval=25, slice_count=n,
                             slice_first=k, slice_num=1) for k in range(n)]
    try:
        for f in range(4):
            y, uv = ol.gen_frame("B", w, h, f)
            ref = whole.encode_frame(y, uv)
            cat = b"".join(p.encode_frame(y, uv) for p in parts)
            assert cat == ref, "frame %d" % f
    finally:
        whole.close()
        for p in parts:
            p.close()


def test_batched_streams_equal_independent_streams(ce):
    """configs[4] structure: several lock-step streams in one session (same kernel
    launches) must each produce exactly what a stream encoded alone produces."""
    w, h, S = 320, 240, 3
    oes = [ol.OracleEncoder(ol.default_config(w, h, qp=24, keyint=25)) for _ in range(S)]
    with ce.CedarEncoder(width=w, height=h, qp=24, keyframe_interval=25, n_streams=S) as ge:
        for f in range(5):
            refs = []
            for s in range(S):
                y, uv = ol.gen_frame("B", w, h, f, seed_offset=s << 8)
                ge.set_input(y, uv, s)
                refs.append(oes[s].encode(y, uv))
            total = ge.encode()
            assert total == sum(len(r) for r in refs)
            for s in range(S):
                assert ge.output(s) == refs[s], "frame %d stream %d" % (f, s)


@pytest.mark.parametrize("batch_lookahead", ["1", "0"])
def test_many_streams_back_to_back_with_idr_lookahead(ce, monkeypatch, batch_lookahead):
    """More than 8 units: frames run back to back, 6 in flight (22 units on the row-per-wave deblock kernel); IDR pictures
    still leave the frame order (streams of their own, events instead of progress words). 11 streams, two slices
    each, keyint 3, the ring kept full: every stream equals its own oracle."""
    monkeypatch.setenv("CEDAR_LOOKAHEAD_BATCH", batch_lookahead)
    w, h, S, n = 176, 144, 11, 10
    frames = [[ol.gen_frame("B", w, h, f, s << 8) for s in range(S)] for f in range(n)]
    oes = [ol.OracleEncoder(ol.default_config(w, h, qp=27, keyint=3, slice_count=2)) for _ in range(S)]
    want = [[oes[s].encode(*frames[f][s]) for s in range(S)] for f in range(n)]
    with ce.CedarEncoder(width=w, height=h, qp=27, keyframe_interval=3, slice_count=2, n_streams=S) as enc:
        assert enc.pipeline_depth == 6 and not enc.stats()["frames_pipelined"]
        got = _encode_pipelined(enc, frames, streams=S)
    assert got == want


@pytest.mark.parametrize("cohorts,streams", [(2, 5), (3, 18)])
def test_cohorts_of_streams_give_the_same_bytes(ce, monkeypatch, cohorts, streams):
    """CEDAR_COHORTS: the streams of a batch go through the stages in groups (the same launches on
    offset pointers, scratch per cohort). Every stream's bytes must be those of a session of its own."""
    w, h = 192, 112
    monkeypatch.setenv("CEDAR_COHORTS", str(cohorts))
    monkeypatch.setenv("CEDAR_ROW_PIPE", "0")
    frames = [[ol.gen_frame("B", w, h, f, s << 8) for s in range(streams)] for f in range(7)]
    with ce.CedarEncoder(width=w, height=h, qp=27, keyframe_interval=3, n_streams=streams) as enc:
        got = _encode_pipelined(enc, frames, streams=streams)
    for s in range(streams):
        oe = ol.OracleEncoder(ol.default_config(w, h, qp=27, keyint=3))
        for f in range(7):
            assert got[f][s] == oe.encode(*frames[f][s]), (s, f)


def test_config4_shape_eight_1080p_streams_in_one_session(ce):
    """BASELINE configs[4] shape on one device: 8 lock-step 1920x1080 streams (seed offset
    s<<8); every stream must equal the oracle's encode of that stream alone."""
    w, h, S = 1920, 1080, 8
    oes = [ol.OracleEncoder(ol.default_config(w, h, qp=24, keyint=25)) for _ in range(S)]
    with ce.CedarEncoder(width=w, height=h, qp=24, keyframe_interval=25, n_streams=S) as ge:
        for f in range(2):
            refs = []
            for s in range(S):
                y, uv = ce.gen_pattern("B", w, h, f, s << 8)
                ge.set_input(y, uv, s)
                refs.append(oes[s].encode(y, uv))
            ge.encode()
            for s in range(S):
                assert ge.output(s) == refs[s], "frame %d stream %d" % (f, s)


def test_pipelined_submit_collect_equals_synchronous(ce):
    w, h = 320, 240
    frames = [ol.gen_frame("B", w, h, f) for f in range(7)]
    with ce.CedarEncoder(width=w, height=h, qp=30, keyframe_interval=3) as a:
        sync = [a.encode_frame(y, uv) for y, uv in frames]
    with ce.CedarEncoder(width=w, height=h, qp=30, keyframe_interval=3) as b:
        out = []
        b.set_input(*frames[0])
        b.submit()
        for f in range(1, len(frames)):
            b.set_input(*frames[f])
            b.submit()           # two frames in flight
            b.collect()
            out.append(b.output())
        b.collect()
        out.append(b.output())
    assert out == sync


def _encode_pipelined(enc, frames, streams=1):
    """Keep as many frames in flight as the session allows; returns the access units in order
    (a list per frame when streams > 1)."""
    depth, out, q = enc.pipeline_depth, [], 0
    for fr in frames:
        if q == depth:
            enc.collect()
            out.append(enc.output() if streams == 1 else [enc.output(s) for s in range(streams)])
            q -= 1
        if streams == 1:
            enc.set_input(*fr)
        else:
            for s in range(streams):
                enc.set_input(fr[s][0], fr[s][1], stream=s)
        enc.submit()
        q += 1
    while q:
        enc.collect()
        out.append(enc.output() if streams == 1 else [enc.output(s) for s in range(streams)])
        q -= 1
    return out


@pytest.mark.parametrize("w,h,qp,keyint,n", [(320, 240, 28, 4, 14), (854, 480, 24, 25, 8), (1920, 1080, 24, 3, 7)])
def test_full_depth_pipeline_matches_oracle(ce, w, h, qp, keyint, n):
    """Frames in flight overlap on the device (frame n+1 is analysed in diagonal stripes behind
    frame n's deblocking wavefront, several deblock kernels run at once): same bytes as the
    scalar oracle, IDR pictures included."""
    frames = [ol.gen_frame("B", w, h, f) for f in range(n)]
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=keyint))
    want = [oe.encode(y, uv) for y, uv in frames]
    with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint) as enc:
        assert enc.pipeline_depth >= 2
        got = _encode_pipelined(enc, frames)
    assert [len(g) for g in got] == [len(x) for x in want]
    assert got == want


def test_frame_pipelining_on_off_and_stripe_counts_agree(ce, monkeypatch):
    """Every schedule the session can be configured into writes the same bytes: frames back to
    back, pipelined with different stripe counts, with and without the IDR look-ahead, shallow and
    deep rings, the IDR picture's deblocking beside or behind its intra wavefront, a third deblock
    stream, a second analysis stream, a second entropy lane."""
    w, h = 640, 368
    frames = [ol.gen_frame("C", w, h, f) for f in range(13)]
    knobs = ("CEDAR_ROW_PIPE", "CEDAR_INTER_BANDS", "CEDAR_ANALYSIS_STREAMS", "CEDAR_LOOKAHEAD", "CEDAR_PIPE_DEPTH",
             "CEDAR_IDR_DEBLOCK_INLINE", "CEDAR_DEBLOCK_STREAMS", "CEDAR_ENTROPY_LANES")
    outs, depths = {}, {}
    for key, env in (("off", {"CEDAR_ROW_PIPE": "0"}), ("on", {"CEDAR_ROW_PIPE": "1"}),
                     ("2 stripes", {"CEDAR_ROW_PIPE": "1", "CEDAR_INTER_BANDS": "2"}),
                     ("11 stripes, 1 analysis stream", {"CEDAR_ROW_PIPE": "1", "CEDAR_INTER_BANDS": "11",
                                                        "CEDAR_ANALYSIS_STREAMS": "1"}),
                     ("no look-ahead, depth 4", {"CEDAR_LOOKAHEAD": "0", "CEDAR_PIPE_DEPTH": "4"}),
                     ("no look-ahead, depth 16", {"CEDAR_LOOKAHEAD": "0"}),
                     ("depth 5", {"CEDAR_PIPE_DEPTH": "5"}),
                     ("idr deblock beside intra", {"CEDAR_IDR_DEBLOCK_INLINE": "0"}),
                     ("3 deblock streams", {"CEDAR_DEBLOCK_STREAMS": "3"}),
                     ("2 analysis streams", {"CEDAR_ANALYSIS_STREAMS": "2"}),
                     ("2 entropy lanes", {"CEDAR_ENTROPY_LANES": "2"})):
        for k in knobs:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)   # read by cedar_enc_config
        with ce.CedarEncoder(width=w, height=h, qp=26, keyframe_interval=4) as enc:
            depths[key] = enc.pipeline_depth
            outs[key] = _encode_pipelined(enc, frames)
    for key in outs:
        assert outs[key] == outs["off"], key
    assert depths["on"] == 16 and depths["off"] == 4 and depths["depth 5"] == 5 and depths["no look-ahead, depth 4"] == 4
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=26, keyint=4))
    assert outs["off"] == [oe.encode(y, uv) for y, uv in frames]


def test_pipelined_multi_slice_and_multi_stream_sessions(ce):
    """Progress words are per unit (stream x slice): a 3-slice picture and a 2-stream session,
    both pipelined, against the oracle."""
    w, h = 640, 480
    frames = [ol.gen_frame("B", w, h, f) for f in range(7)]
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=27, keyint=5, slice_count=3))
    want = [oe.encode(y, uv) for y, uv in frames]
    with ce.CedarEncoder(width=w, height=h, qp=27, keyframe_interval=5, slice_count=3) as enc:
        assert _encode_pipelined(enc, frames) == want
    two = [[ol.gen_frame("B", w, h, f, seed_offset=s << 8) for s in range(2)] for f in range(6)]
    want2 = []
    oes = [ol.OracleEncoder(ol.default_config(w, h, qp=27, keyint=5)) for _ in range(2)]
    for fr in two:
        want2.append([oes[s].encode(fr[s][0], fr[s][1]) for s in range(2)])
    with ce.CedarEncoder(width=w, height=h, qp=27, keyframe_interval=5, n_streams=2) as enc:
        assert _encode_pipelined(enc, two, streams=2) == want2


def test_device_resident_input_equals_host_input(ce):
    torch = pytest.importorskip("torch")
    w, h = 320, 240
    with ce.CedarEncoder(width=w, height=h, qp=24, keyframe_interval=25) as a, \
            ce.CedarEncoder(width=w, height=h, qp=24, keyframe_interval=25) as b:
        for f in range(3):
            y, uv = ol.gen_frame("B", w, h, f)
            ref = a.encode_frame(y, uv)
            ty = torch.from_numpy(y).cuda()
            tuv = torch.from_numpy(uv).cuda()
            torch.cuda.synchronize()
            b.set_device_input(0, ty.data_ptr(), tuv.data_ptr())
            b.encode()
            assert b.output() == ref


def test_repeat_headers_and_gop_sharding(ce):
    """Closed GOPs shard: with repeat_headers every GOP starts with SPS+PPS+IDR and
    depends on nothing before it, so a second session that starts at GOP 1 (fresh state, frames
    25..) produces GOP 1 of the serial encode except for idr_pic_id / frame counters it cannot
    know -- here both sessions encode from their frame 0, so the bytes must be identical."""
    w, h = 320, 240
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=30, keyint=3, repeat_headers=1))
    with ce.CedarEncoder(width=w, height=h, qp=30, keyframe_interval=3, repeat_headers=1) as ge:
        for f in range(7):
            y, uv = ol.gen_frame("B", w, h, f)
            ref = oe.encode(y, uv)
            got = ge.encode_frame(y, uv)
            assert got == ref
            kinds = [got[i + 4] & 0x1f for i in range(len(got) - 4) if got[i:i + 4] == b"\0\0\0\1"]
            assert kinds == ([7, 8, 5] if f % 3 == 0 else [1])


def test_access_unit_delimiters_and_vui_timing(ce):
    """SURVEY 8(f)-3, header extensions beside the path: an access unit delimiter opens every
    access unit and the SPS carries VUI timing. Same bytes as the oracle (synchronous, pipelined,
    two lock-step streams), the independent decoder reads the frame rate back, and with slices
    sharded over sessions only the first slice's session writes the delimiter and parameter sets."""
    w, h, n = 320, 240, 9
    kw = dict(aud=1, fps_num=30000, fps_den=1001, repeat_headers=1)
    frames = [ol.gen_frame("B", w, h, f) for f in range(n)]
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=26, keyint=4, **kw))
    want = [oe.encode(y, uv) for y, uv in frames]
    assert want[0][:6] == bytes([0, 0, 0, 1, 9, 0x10]) and want[1][:6] == bytes([0, 0, 0, 1, 9, 0x30])
    with ce.CedarEncoder(width=w, height=h, qp=26, keyframe_interval=4, **kw) as enc:
        assert [enc.encode_frame(y, uv) for y, uv in frames] == want
    with ce.CedarEncoder(width=w, height=h, qp=26, keyframe_interval=4, **kw) as enc:
        assert _encode_pipelined(enc, frames) == want
    dec = ol.OracleDecoder()
    for au in want:
        assert dec.decode(au) == 1
    assert dec.frame_rate() == (60000, 2002) and dec.aud_count() == n and dec.reorder_frames() == 0
    # two streams in one session: each gets its own delimiter + parameter sets
    fr2 = [[ol.gen_frame("B", w, h, f, s << 8) for s in range(2)] for f in range(5)]
    oes = [ol.OracleEncoder(ol.default_config(w, h, qp=26, keyint=4, **kw)) for _ in range(2)]
    want2 = [[oes[s].encode(*fr2[f][s]) for s in range(2)] for f in range(5)]
    with ce.CedarEncoder(width=w, height=h, qp=26, keyframe_interval=4, n_streams=2, **kw) as enc:
        assert _encode_pipelined(enc, fr2, streams=2) == want2
    # slices on separate sessions: the concatenation is the whole, so only slice 0 carries the front matter
    oe3 = ol.OracleEncoder(ol.default_config(w, h, qp=26, keyint=4, slice_count=3, **kw))
    parts = [ce.CedarEncoder(width=w, height=h, qp=26, keyframe_interval=4, slice_count=3, slice_first=k,
                             slice_num=1, **kw) for k in range(3)]
    try:
        for f in range(5):
            assert b"".join(p.encode_frame(*frames[f]) for p in parts) == oe3.encode(*frames[f]), f
    finally:
        for p in parts:
            p.close()
    for bad in (dict(fps_num=30), dict(fps_den=1), dict(aud=2), dict(fps_num=1 << 30, fps_den=1)):
        with pytest.raises(ce.EncoderError) as ex:
            ce.CedarEncoder(width=w, height=h, **bad)
        assert ex.value.errno == errno.EINVAL, bad


@pytest.mark.parametrize("off", [-12, -3, 0, 12])
def test_chroma_qp_offset_variants(ce, off):
    """chroma_qp_index_offset other than the reference's +4 (cedar.c:971-972, :1165-1168): chroma
    quantiser, the chroma deblock thresholds and the PPS follow it; decoder round trip included."""
    w, h = 208, 112
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=29, keyint=3, chroma_qp_offset=off))
    dec = ol.OracleDecoder()
    with ce.CedarEncoder(width=w, height=h, qp=29, keyframe_interval=3, chroma_qp_offset_minus4=off - 4) as ge:
        for f in range(5):
            y, uv = ol.gen_frame("B", w, h, f)
            ref = oe.encode(y, uv)
            assert ge.encode_frame(y, uv) == ref, f
            _compare_state(oe, ge)
            assert dec.decode(ref) == 1
            dy, duv = dec.picture()
            assert np.array_equal(dy, ge.debug_read("recon_y")) and np.array_equal(duv, ge.debug_read("recon_uv"))
    for bad in (-17, 9):
        with pytest.raises(ce.EncoderError) as ex:
            ce.CedarEncoder(width=w, height=h, chroma_qp_offset_minus4=bad)
        assert ex.value.errno == errno.EINVAL


def test_session_contract_errors(ce):
    """Error behaviour of the reference's ioctls (cedar.c:738-789, :1039-1043)."""
    e = ce.CedarEncoder()
    with pytest.raises(ce.EncoderError) as ex:  # ENCODE before CONFIG -> EINVAL
        e.encode()
    assert ex.value.errno == errno.EINVAL
    for bad in (dict(width=321, height=240), dict(width=320, height=240, qp=0),
                dict(width=320, height=240, qp=48), dict(width=320, height=240, keyframe_interval=0),
                dict(width=320, height=240, keyframe_interval=32),
                dict(width=320, height=240, entropy_coding_mode=1),   # CABAC: rejected, never silently CAVLC
                dict(width=320, height=240, src_format=1),
                dict(width=320, height=240, dst_width=336)):
        with pytest.raises(ce.EncoderError) as ex:
            e.config(ce.EncoderConfig(**bad))
        assert ex.value.errno == errno.EINVAL, bad
    e.config(ce.EncoderConfig(width=320, height=240))
    with pytest.raises(ce.EncoderError) as ex:  # second CONFIG -> EINVAL (cedar.c:744-747)
        e.config(ce.EncoderConfig(width=320, height=240))
    assert ex.value.errno == errno.EINVAL
    e.close()


def test_gop_cadence_headers_and_sizes(ce):
    """SPS+PPS only in front of frame 0 (cedar.c:1058-1061); I P P I ... by keyint;
    encode() returns the byte count valid at bytestream offset 0 (cedar.c:1207)."""
    with ce.CedarEncoder(width=64, height=48, qp=30, keyframe_interval=3) as e:
        y, uv = ol.gen_frame("B", 64, 48, 0)
        kinds = []
        for f in range(7):
            e.set_input(y, uv)
            n = e.encode()
            au = e.output()
            assert n == len(au)
            kinds.append([au[i + 4] & 0x1f for i in range(len(au) - 4) if au[i:i + 4] == b"\0\0\0\1"])
        assert kinds[0] == [7, 8, 5]
        assert kinds[1:] == [[1], [1], [5], [1], [1], [5]]


def test_two_sessions_interleaved_pipelined(ce):
    """Only one session per device pipelines its frames on the device (two would queue polling
    kernels behind each other); a second one still accepts frames in flight and both deliver
    the oracle's bytes when driven alternately."""
    w, h = 320, 240
    frames = [ol.gen_frame("B", w, h, f) for f in range(9)]
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=26, keyint=4))
    want = [oe.encode(y, uv) for y, uv in frames]
    with ce.CedarEncoder(width=w, height=h, qp=26, keyframe_interval=4) as a, \
            ce.CedarEncoder(width=w, height=h, qp=26, keyframe_interval=4) as b:
        outs = {id(a): [], id(b): []}
        depth = min(a.pipeline_depth, b.pipeline_depth)
        q = 0
        for fr in frames:
            if q == depth:
                for enc in (a, b):
                    enc.collect()
                    outs[id(enc)].append(enc.output())
                q -= 1
            for enc in (a, b):
                enc.set_input(*fr)
                enc.submit()
            q += 1
        while q:
            for enc in (a, b):
                enc.collect()
                outs[id(enc)].append(enc.output())
            q -= 1
    assert outs[id(a)] == want
    assert outs[id(b)] == want


@pytest.mark.parametrize("env", [
    {"CEDAR_WF_SPLIT": "1", "CEDAR_WF_SPLIT_INTRA": "1"},            # 17 bands on one workgroup: bands wrap around its wave teams
    {"CEDAR_WF_SPLIT": "3", "CEDAR_WF_SPLIT_INTRA": "2", "CEDAR_DEBLOCK_PLANES": "0"},  # both planes in every wave pair
    {"CEDAR_WF_SPLIT": "8", "CEDAR_WF_SPLIT_INTRA": "8", "CEDAR_ROW_PIPE": "0"},
    {"CEDAR_ENTROPY_FUSED": "0", "CEDAR_INTRA_ROWS": "0"},   # round 3's back end (k_zero .. k_esc_write, k_copy_out) and intra wavefront
    {"CEDAR_STREAM_PRIO": "1"},                               # the analysis stream at the greatest priority, roles dealt in classes
    {"CEDAR_TAIL_GATE": "0"},                                 # a k_gate in front of every stripe and the held-back tail, as until round 4
])
def test_1080p_wavefront_shapes_match_oracle(ce, monkeypatch, env):
    """The wavefront kernels' work split (workgroups per unit, bands per workgroup, planes per wave
    pair) is a launch-shape choice: every shape must give the oracle's bytes."""
    for k in ("CEDAR_WF_SPLIT", "CEDAR_WF_SPLIT_INTRA", "CEDAR_DEBLOCK_PLANES", "CEDAR_ROW_PIPE", "CEDAR_ENTROPY_FUSED", "CEDAR_INTRA_ROWS", "CEDAR_STREAM_PRIO", "CEDAR_TAIL_GATE"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    w, h = 1920, 1080
    frames = [ol.gen_frame("B", w, h, f) for f in range(3)]
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=26, keyint=25))
    want = [oe.encode(y, uv) for y, uv in frames]
    with ce.CedarEncoder(width=w, height=h, qp=26, keyframe_interval=25) as enc:
        got = _encode_pipelined(enc, frames)
    assert got == want


@pytest.mark.parametrize("fused", ["1", "0", "1+write-through"])
def test_both_entropy_back_ends_give_the_oracles_bytes(ce, monkeypatch, fused):
    """k_pack_out (one launch behind k_scan: bits, emulation prevention, NAL layout, bytes on the host) and the chain it
    replaces (CEDAR_ENTROPY_FUSED=0) on what stresses a back end: slices and lock-step streams (many NAL units to lay out),
    I_PCM pictures (byte-aligned raw samples, long zero runs to escape: a black picture), pictures of a few bytes."""
    monkeypatch.setenv("CEDAR_ENTROPY_FUSED", fused[0])
    if "write-through" in fused:   # k_pack_out's other way to the host: write-through stores, the flag from its last workgroup
        monkeypatch.setenv("CEDAR_PK_DELIVER", "1")
    rng = np.random.RandomState(7)
    cases = [(320, 240, 24, 3, 3, "B"), (640, 368, 1, 1, 1, "noise"), (96, 64, 4, 2, 2, "black"), (1920, 1080, 40, 1, 1, "static")]
    for w, h, qp, slices, streams, kind in cases:
        def frame(f, s):
            if kind == "noise":
                r = np.random.RandomState(100 * f + s)
                return r.randint(0, 256, w * h).astype(np.uint8), r.randint(0, 256, w * h // 2).astype(np.uint8)
            if kind == "black":   # I_PCM cannot happen here; zero samples with a tiny texture: long zero runs in a short stream
                y = np.zeros(w * h, np.uint8)
                y[(f * 7 + s) % 50::97] = 3
                return y, np.zeros(w * h // 2, np.uint8)
            if kind == "static":
                return ol.gen_frame("A", w, h, 0)
            return ol.gen_frame(kind, w, h, f, s << 8)
        n = 4
        oes = [ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=3, slice_count=slices)) for _ in range(streams)]
        with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=3, slice_count=slices, n_streams=streams) as enc:
            for f in range(n):
                for s in range(streams):
                    enc.set_input(*frame(f, s), stream=s)
                enc.encode()
                for s in range(streams):
                    assert enc.output(s) == oes[s].encode(*frame(f, s)), (w, h, kind, f, s)
    del rng


def test_long_pipelined_run_many_gops(ce):
    """120 frames, 8 GOPs, everything in flight that the session allows: every access unit equals
    the oracle's (progress words are tagged per frame and never cleared; slots, streams and
    rings are reused dozens of times)."""
    w, h = 640, 368
    n = 120
    frames = [ol.gen_frame("BC"[f % 2], w, h, f) for f in range(n)]
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=29, keyint=15))
    want = [oe.encode(y, uv) for y, uv in frames]
    with ce.CedarEncoder(width=w, height=h, qp=29, keyframe_interval=15) as enc:
        got = _encode_pipelined(enc, frames)
    bad = [i for i in range(n) if got[i] != want[i]]
    assert not bad, "first differing frames: %s" % bad[:8]


@pytest.mark.parametrize("which,keyint", [(1, 25), (2, 1)], ids=["deblock-mover", "intra-mover"])
def test_lost_handoff_drains_and_reports_eio(ce, monkeypatch, which, keyint):
    """Failure detection: every hand-off wait in the wavefront kernels is bounded, and the first
    one that runs out raises abort flags so that the rest of the grid (and the frames queued behind
    it) drains at once instead of timing out wait by wait. CEDAR_FAULT_INJECT makes one mover wave
    stop reporting: the encode call must come back with -EIO within seconds, GOP state must not
    advance (cedar.c:1182-1191), and the next session must work as if nothing had happened."""
    import time
    w, h = 640, 368
    frames = [ol.gen_frame("B", w, h, f) for f in range(3)]
    monkeypatch.setenv("CEDAR_FAULT_INJECT", str(which))
    with ce.CedarEncoder(width=w, height=h, qp=26, keyframe_interval=keyint) as bad:
        t0 = time.time()
        with pytest.raises(ce.EncoderError) as ex:
            for y, uv in frames:
                bad.encode_frame(y, uv)
        assert ex.value.errno == errno.EIO
        assert time.time() - t0 < 60
    monkeypatch.delenv("CEDAR_FAULT_INJECT")
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=26, keyint=keyint))
    with ce.CedarEncoder(width=w, height=h, qp=26, keyframe_interval=keyint) as good:
        for y, uv in frames:
            assert good.encode_frame(y, uv) == oe.encode(y, uv)


def test_hip_streams_equal_the_fullsize_golden_vectors(ce):
    """tests/golden/indep_fullsize.json (tools/indep_fullsize.py): the first GOP of BASELINE configs[1], [2] and [3]
    as the oracle emits it, every picture rebuilt by the independent Python decoder and checked against the level
    limits. Here the HIP path's access units are only hashed against those vectors: at the sizes the metric is
    quoted on, its bytes are pinned to a reading of the standard that shares nothing with oracle/ or csrc/."""
    import hashlib
    import json
    cases = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "indep_fullsize.json")))["cases"]
    assert [c["case"][0] for c in cases] == ["configs[1]", "configs[2]", "configs[3]"] + ["configs[4] stream %d" % s for s in range(1, 8)]
    for c in cases[:3]:
        name, w, h, qp, keyint, frames, slices, fps = c["case"]
        with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint, slice_count=slices, level=c["level_idc"],
                             fps_num=fps, fps_den=1) as enc:
            got = _encode_pipelined(enc, [ol.gen_frame("B", w, h, f) for f in range(frames)])
        assert [hashlib.sha1(au).hexdigest() for au in got] == c["au_sha1"], name
        assert hashlib.sha1(b"".join(got)).hexdigest() == c["stream_sha1"], name
    # configs[4]: the 8 concurrent 1080p streams as ONE lock-step session; stream 0 is configs[2]'s, stream s is fed
    # pattern-B with seed offset s << 8 (what bench.py feeds it), the first 3 access units of each
    name, w, h, qp, keyint, _, _, fps = cases[1]["case"]
    with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint, n_streams=8, level=41, fps_num=fps, fps_den=1) as enc:
        got = [[] for _ in range(8)]
        for f in range(3):
            for s in range(8):
                enc.set_input(*ol.gen_frame("B", w, h, f, s << 8), stream=s)
            enc.encode()
            for s in range(8):
                got[s].append(hashlib.sha1(enc.output(s)).hexdigest())
    assert got[0] == cases[1]["au_sha1"][:3]
    for s in range(1, 8):
        assert cases[2 + s]["case"][8] == s << 8 and got[s] == cases[2 + s]["au_sha1"], s


def test_hip_streams_equal_the_independent_decoders_golden_vectors(ce):
    """tests/golden/indep_decode.json (tools/gen_indep_golden.py): the bytes the HIP path emits for
    the committed cases must hash to the committed stream vectors, and the independent Python decoder
    (tests/indep/h264_refdec.py: own code and tables, nothing shared with oracle/ or csrc/) must turn
    them into the session's own deblocked reconstruction."""
    import hashlib
    import json
    from tests.indep import h264_refdec as rd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = json.load(open(os.path.join(root, "tests", "golden", "indep_decode.json")))["cases"]
    for c in cases:
        w, h, qp, keyint, frames, pattern, slices = c["case"][:7]
        sys.path.insert(0, os.path.join(root, "tools"))
        from gen_indep_golden import case_options
        opts = case_options(c["case"])
        sh, dec = hashlib.sha1(), rd.Decoder()
        with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint, slice_count=slices, **opts) as ge:
            for f in range(frames):
                au = ge.encode_frame(*ol.gen_frame(pattern, w, h, f))
                sh.update(au)
                pic, = dec.decode(au)
                assert pic.luma_bytes() == ge.debug_read("recon_y").tobytes(), (c["case"], f)
                assert pic.nv12_chroma_bytes() == ge.debug_read("recon_uv").tobytes(), (c["case"], f)
        assert sh.hexdigest() == c["stream_sha1"], c["case"]


def _drifting_texture(w, h, f, vx=1.25, vy=0.5):
    """a smooth texture that moves by a fraction of a sample per frame: what sub-sample vectors are for"""
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    u, v = x + vx * f, y + vy * f
    lum = 128 + 60 * np.sin(u / 5.0) * np.cos(v / 7.0) + 30 * np.sin((u + v) / 3.0)
    yy = np.clip(np.round(lum), 0, 255).astype(np.uint8).reshape(-1)
    cy, cx = np.mgrid[0:h // 2, 0:w // 2].astype(np.float64)
    uv = np.empty((h // 2, w), np.uint8)
    uv[:, 0::2] = np.clip(np.round(128 + 40 * np.sin((cx * 2 + vx * f) / 9.0)), 0, 255).astype(np.uint8)
    uv[:, 1::2] = np.clip(np.round(128 + 40 * np.cos((cy * 2 + vy * f) / 11.0)), 0, 255).astype(np.uint8)
    return yy, uv.reshape(-1)


@pytest.mark.parametrize("w,h,qp,subpel,slices,search", [(96, 80, 26, 1, 1, 0), (96, 80, 26, 2, 1, 0), (320, 240, 30, 2, 3, 0),
                                                         (70, 50, 20, 2, 1, 5), (640, 368, 24, 2, 1, 0), (1920, 1080, 24, 2, 1, 0)])
def test_sub_sample_motion_vectors_equal_the_oracle(ce, w, h, qp, subpel, slices, search):
    """subpel = 1 / 2: half- and quarter-sample refinement around the best whole-sample vector (6-tap
    planes built per macroblock in LDS). Decisions, levels, reconstruction and bytes equal the scalar
    twin's; fractional vectors really occur; the independent decoder interpolates the same pictures."""
    from tests.indep import h264_refdec as rd
    n = 4 if w < 1000 else 2
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=25, subpel=subpel, slice_count=slices, search_range=search or 16))
    dec = rd.Decoder() if w <= 320 else None
    fr = 0
    with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=25, subpel=subpel, slice_count=slices, search_range=search) as ge:
        for f in range(n):
            y, uv = _drifting_texture(w, h, f) if f < 3 else ol.gen_frame("B", w, h, f)
            ref = oe.encode(y, uv)
            got = ge.encode_frame(y, uv)
            _compare_state(oe, ge)
            assert got == ref, "frame %d: bytestream differs" % f
            mb = ge.debug_read("mbinfo")
            fr += int((((mb["mvx"] | mb["mvy"]) & 3) != 0).sum())
            if dec:
                pic, = dec.decode(got)
                assert pic.luma_bytes() == ge.debug_read("recon_y").tobytes()
    assert fr > 0
    if dec:
        assert dec.cover["mc_luma_fractional"] > 0


@pytest.mark.parametrize("w,h,target", [(320, 240, 500000), (854, 480, 2500000)])
def test_rate_control_follows_the_oracle_at_any_depth(ce, w, h, target):
    """bitrate > 0: the picture QP comes from sizes of frames 17 and more back (collected by then
    whatever the pipelining), so the synchronous loop, the full ring and the scalar oracle pick the
    same QPs and emit the same bytes."""
    n = 48
    frames = [ol.gen_frame("B", w, h, f) for f in range(n)]
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=30, keyint=25, bitrate=target))
    want, qps = [], []
    for y, uv in frames:
        want.append(oe.encode(y, uv))
        qps.append(oe.frame_qp())
    assert len(set(qps)) > 1
    with ce.CedarEncoder(width=w, height=h, qp=30, keyframe_interval=25, bitrate=target) as enc:
        got = _encode_pipelined(enc, frames)
        assert enc.stats()["frame_qp"] == qps[-1]
    assert got == want
    with ce.CedarEncoder(width=w, height=h, qp=30, keyframe_interval=25, bitrate=target) as enc:
        assert [enc.encode_frame(y, uv) for y, uv in frames[:24]] == want[:24]
    with pytest.raises(ce.EncoderError):     # GOP lanes have a law of their own (below), for one stream
        ce.CedarEncoder(width=w, height=h, qp=30, keyframe_interval=25, bitrate=target, gop_parallel=2, n_streams=2)


def test_gop_lanes_first_submit_does_not_race_the_ring_allocation(ce, monkeypatch):
    """A gop_parallel session fed through the pinned buffers allocates its device-side frame ring at the first submit; the
    allocator's hipMemset runs on the null stream and the copy of that very frame on a non-blocking one, so the clear
    could land on top of frame 0 (round 3's fuzz, seed 303 case 72: 1 session in ~1000; without IDR look-ahead 9 in 10).
    Thirty short sessions of the kind that showed it, look-ahead off: every access unit the oracle's."""
    monkeypatch.setenv("CEDAR_LOOKAHEAD", "0")
    w, h, K, G = 32, 16, 5, 6
    kw = dict(subpel=1, mad_th=[1, 5, 10, 12], mad_dqp=[0, -9, -10, 3, 3])
    for it in range(30):
        frames = [ol.gen_frame("C", w, h, f, 31 * it) for f in range(8)]
        oe = ol.OracleEncoder(ol.default_config(w, h, qp=1, keyint=K, **kw))
        want = [oe.encode(y, uv) for y, uv in frames]
        with ce.CedarEncoder(width=w, height=h, qp=1, keyframe_interval=K, gop_parallel=G, **kw) as enc:
            assert _encode_gop_parallel(enc, frames) == want, it


@pytest.mark.parametrize("w,h,keyint,G,target,n", [(320, 240, 6, 2, 5480000, 100), (192, 112, 4, 3, 1480000, 120), (640, 368, 25, 2, 16750000, 160)])
def test_rate_control_with_gop_lanes_is_a_law_per_group(ce, w, h, keyint, G, target, n):
    """bitrate x gop_parallel (one stream): the later GOPs of a group of G are coded beside the earlier ones, so nothing
    inside a group can steer it -- one QP per group, from the mean bits per frame of the group 1 + ceil(16 / keyint)
    groups back (every step of which has been collected when this group's first step is submitted, whatever the
    pipelining). The scalar twin runs the same law serially (h264o_config.rc_lanes): same QPs, same bytes, and the
    stream still leaves in frame order."""
    frames = [ol.gen_frame("BC"[(f // 7) % 2], w, h, f) for f in range(n)]
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=30, keyint=keyint, bitrate=target, rc_lanes=G))
    want, qps = [], []
    for y, uv in frames:
        want.append(oe.encode(y, uv))
        qps.append(oe.frame_qp())
    lag = 1 + (16 + keyint - 1) // keyint
    assert set(qps[:lag * G * keyint]) == {30} and len(set(qps)) > 1           # the law sets in after `lag` groups ...
    assert all(len(set(qps[i:i + G * keyint])) == 1 for i in range(0, n, G * keyint))   # ... and a group has one QP
    with ce.CedarEncoder(width=w, height=h, qp=30, keyframe_interval=keyint, bitrate=target, gop_parallel=G) as enc:
        got = _encode_gop_parallel(enc, frames)
    assert got == want
    dec = ol.OracleDecoder()
    for au in got:
        assert dec.decode(au) == 1


@pytest.mark.parametrize("w,h,S,target,extra", [(320, 240, 3, 500000, {}), (192, 112, 9, 150000, {}),
                                                (320, 240, 2, 700000, dict(roi=((2, 2, 8, 6, -6),), subpel=1))])
def test_rate_control_over_several_streams(ce, w, h, S, target, extra):
    """bitrate > 0 with n_streams > 1: the law runs per stream (its own sizes, its own QP per picture, in its own
    slice_qp_delta); the kernels take the picture's QP per stream where they otherwise take the launch's
    (Params::qp_s, the per-macroblock-QP forms). Every stream equals the one-stream oracle with the same target."""
    n = 44
    pats = "BCA"
    src = [[ol.gen_frame(pats[s % 3], w, h, f, s << 8) for s in range(S)] for f in range(n)]
    okw = {k: (list(v) if k == "roi" else v) for k, v in extra.items()}
    want, qps = [], []
    for s in range(S):
        oe = ol.OracleEncoder(ol.default_config(w, h, qp=30, keyint=25, bitrate=target, pcm=-1, **okw))   # (several steered streams: no I_PCM)
        aus, q = [], []
        for f in range(n):
            aus.append(oe.encode(*src[f][s]))
            q.append(oe.frame_qp())
        want.append(aus)
        qps.append(q)
    assert len({tuple(q) for q in qps}) > 1      # the streams really go different ways
    with ce.CedarEncoder(width=w, height=h, qp=30, keyframe_interval=25, bitrate=target, n_streams=S, **extra) as enc:
        depth, got, q = enc.pipeline_depth, [], 0
        for f in range(n):
            if q == depth:
                enc.collect(); got.append([enc.output(s) for s in range(S)]); q -= 1
            for s in range(S):
                enc.set_input(src[f][s][0], src[f][s][1], stream=s)
            enc.submit(); q += 1
        while q:
            enc.collect(); got.append([enc.output(s) for s in range(S)]); q -= 1
        assert enc.stats()["frame_qp"] == qps[0][-1]
    for s in range(S):
        bad = [f for f in range(n) if got[f][s] != want[s][f]]
        assert not bad, (s, bad[:5])


def test_macroblocks_above_3200_bits_are_counted_or_refused(ce):
    """H.264 A.3.1: a macroblock_layer() may take 3200 bits. Pure noise at a low QP passes that (measured
    with the independent decoder: up to QP 16) where I_PCM is switched off (pcm = -1) or not taken. The path detects
    it: by default the picture is emitted and counted, with strict_mb_bits it fails with EOVERFLOW and
    is dropped like any failed frame (GOP state untouched) -- the oracle does the same."""
    from tests.indep import h264_refdec as rd
    w, h = 64, 48
    frames = [ol.gen_frame("C", w, h, f) for f in range(3)] + [ol.gen_frame("A", w, h, 3)]
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=1, keyint=1, pcm=-1))
    dec = rd.Decoder()
    with ce.CedarEncoder(width=w, height=h, qp=1, keyframe_interval=1, pcm=-1) as enc:
        for y, uv in frames:
            au = enc.encode_frame(y, uv)
            assert au == oe.encode(y, uv)
            dec.decode(au)
        assert dec.max_mb_bits > 3200 and oe.max_mb_bits() <= 3200   # the noise frames pass it, the flat last one does not
        assert enc.stats()["oversize_frames"] == 3
    os_ = ol.OracleEncoder(ol.default_config(w, h, qp=1, keyint=1, strict_mb_bits=1, pcm=-1))
    with ce.CedarEncoder(width=w, height=h, qp=1, keyframe_interval=1, strict_mb_bits=1, pcm=-1) as enc:
        for y, uv in frames[:3]:
            with pytest.raises(ce.EncoderError) as ex:
                enc.encode_frame(y, uv)
            assert ex.value.errno == errno.EOVERFLOW
            with pytest.raises(OSError) as ex2:
                os_.encode(y, uv)
            assert ex2.value.errno == errno.EOVERFLOW
        au = enc.encode_frame(*frames[3])              # still frame 0 of the stream: SPS + PPS + IDR
        assert au == os_.encode(*frames[3])
        d2 = rd.Decoder()
        d2.decode(au)
        assert d2.max_mb_bits <= 3200 and d2.pictures == 1
    # at the QPs of the BASELINE configs not even noise comes near the limit
    dq = rd.Decoder()
    with ce.CedarEncoder(width=w, height=h, qp=24, keyframe_interval=1, strict_mb_bits=1, pcm=-1) as enc:
        for y, uv in frames[:3]:
            dq.decode(enc.encode_frame(y, uv))
    assert 0 < dq.max_mb_bits <= 3200


@pytest.mark.parametrize("qp,keyint", [(1, 25), (12, 3)])
def test_oversize_pictures_are_requantised_until_every_macroblock_fits(ce, qp, keyint):
    """strict_mb_bits = 2, the hard cap: a picture with a macroblock above 3200 bits is coded again 6 QP
    steps coarser (slice_qp_delta), with the frames in flight behind it re-encoded on the new
    reconstruction. Same QPs and bytes as the scalar twin, synchronous or with the ring full; the
    independent decoder finds no macroblock above the limit and the session's own reconstruction."""
    from tests.indep import h264_refdec as rd
    w, h, n = 64, 48, 7
    frames = [ol.gen_frame("CB"[f % 5 == 4], w, h, f) for f in range(n)]
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=keyint, strict_mb_bits=2, pcm=-1))
    want, qps = [], []
    for y, uv in frames:
        want.append(oe.encode(y, uv))
        qps.append(oe.frame_qp())
    assert max(qps) > qp
    dec = rd.Decoder()
    with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint, strict_mb_bits=2, pcm=-1) as enc:
        for f, (y, uv) in enumerate(frames):
            au = enc.encode_frame(y, uv)
            assert au == want[f], f
            assert enc.stats()["frame_qp"] == qps[f]
            pic, = dec.decode(au)
            assert pic.luma_bytes() == enc.debug_read("recon_y").tobytes()
        assert enc.stats()["requantised_frames"] > 0 and dec.max_mb_bits <= 3200
    with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint, strict_mb_bits=2, pcm=-1) as enc:
        assert _encode_pipelined(enc, frames) == want
    # the ring really full: more frames than the session takes in flight, all of its depth in flight
    n2 = ce.load_library().cedar_enc_pipeline_depth() + 9
    frames2 = [ol.gen_frame("CB"[f % 5 == 4], w, h, f) for f in range(n2)]
    oe2 = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=keyint, strict_mb_bits=2, pcm=-1))
    want2 = [oe2.encode(y, uv) for y, uv in frames2]
    with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint, strict_mb_bits=2, pcm=-1) as enc:
        assert _encode_pipelined(enc, frames2) == want2
        assert enc.stats()["requantised_frames"] > 0


def test_failed_encode_is_retryable_on_the_same_session(ce, monkeypatch):
    """cedar.c:1182-1191 return before the GOP counter and the reference swap (:1193-1201): the caller
    may simply call ENCODE again. Here the failed attempt leaves completion words, progress words and
    abort flags behind in the very slots the retry uses; they are tagged by a submit sequence that is
    never rolled back, so the retry must neither see "done" early nor a finished analysis."""
    w, h = 640, 368
    frames = [ol.gen_frame("B", w, h, f) for f in range(4)]
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=26, keyint=25))
    want = [oe.encode(y, uv) for y, uv in frames]
    monkeypatch.setenv("CEDAR_FAULT_INJECT", "1")
    monkeypatch.setenv("CEDAR_FAULT_INJECT_AT", "2")   # the second submit (frame 1, a P picture) loses a hand-off
    with ce.CedarEncoder(width=w, height=h, qp=26, keyframe_interval=25) as enc:
        assert enc.encode_frame(*frames[0]) == want[0]
        with pytest.raises(ce.EncoderError) as ex:
            enc.encode_frame(*frames[1])
        assert ex.value.errno == errno.EIO
        for f in (1, 2, 3):                            # the same frame again, then on as if nothing had happened
            assert enc.encode_frame(*frames[f]) == want[f], f
            ry, ruv = enc.debug_read("recon_y"), enc.debug_read("recon_uv")
        dec = ol.OracleDecoder()
        for au in want:
            dec.decode(au)
        dy, duv = dec.picture()
        assert np.array_equal(dy, ry) and np.array_equal(duv, ruv)


@pytest.mark.parametrize("at,keyint,full", [(3, 25, False), (1, 4, False), (5, 4, False), (20, 25, True), (6, 7, True)],
                         ids=["p-picture", "first-idr", "later-idr", "ring-full", "ring-full-short-gops"])
def test_pipelined_failure_drops_exactly_that_frame(ce, monkeypatch, at, keyint, full):
    """The reference's loop goes on with the next frame after a failed ENCODE, with the GOP counter
    and the reference frames untouched (h264enc.c:189-192, cedar.c:1182-1201). With frames in flight
    the session does the same: the failed collect drops that frame, the frames submitted behind it
    are coded again on the state from before it. The stream equals the oracle's encode of the
    sequence WITHOUT the failed frame. "ring-full": as many frames in flight as the session takes
    (cedar_enc_session_depth) and more frames than that in all, so that the frame depth - 1 behind the
    failed one has reused the device slot of the frame before it."""
    w, h = 640, 368
    n = 9 if not full else at + ce.load_library().cedar_enc_pipeline_depth() + 6
    frames = [ol.gen_frame("B", w, h, f) for f in range(n)]
    monkeypatch.setenv("CEDAR_FAULT_INJECT", "1" if keyint in (25, 7) else "2" if at in (1, 5) else "1")
    monkeypatch.setenv("CEDAR_FAULT_INJECT_AT", str(at))
    got, failed = [], []
    with ce.CedarEncoder(width=w, height=h, qp=26, keyframe_interval=keyint) as enc:
        depth, q, nxt = enc.pipeline_depth if full else min(enc.pipeline_depth, 6), 0, 0
        def take():
            nonlocal nxt
            try:
                enc.collect()
                got.append(enc.output())
            except ce.EncoderError as ex:
                assert ex.errno == errno.EIO
                failed.append(nxt)
            nxt += 1
        for f in range(n):
            if q == depth:
                take()
                q -= 1
            enc.set_input(*frames[f])
            enc.submit()
            q += 1
        while q:
            take()
            q -= 1
    assert failed == [at - 1]
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=26, keyint=keyint))
    want = [oe.encode(*frames[f]) for f in range(n) if f != at - 1]
    assert got == want


def test_closed_gop_shards_concatenate_to_the_serial_stream(ce):
    """GOP g on session (GPU) g mod N. idr_pic_id and "SPS + PPS in front of
    frame 0 only" derive from the global frame index (first_frame_index), so the shards' bytes,
    concatenated in GOP order, ARE the serial stream of the oracle -- no repeat_headers needed."""
    w, h, K, gops = 320, 240, 5, 6
    frames = [ol.gen_frame("B", w, h, f) for f in range(K * gops)]
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=28, keyint=K))
    want = b"".join(oe.encode(y, uv) for y, uv in frames)
    n_shards = 4
    parts = {}
    for shard in range(n_shards):
        for g in range(shard, gops, n_shards):       # every GOP is a fresh session start: nothing carries over
            with ce.CedarEncoder(width=w, height=h, qp=28, keyframe_interval=K, first_frame_index=g * K) as enc:
                parts[g] = b"".join(enc.encode_frame(*frames[g * K + t]) for t in range(K))
    assert b"".join(parts[g] for g in range(gops)) == want
    with pytest.raises(ce.EncoderError) as ex:       # a shard starts with a GOP
        ce.CedarEncoder(width=w, height=h, qp=28, keyframe_interval=K, first_frame_index=K + 1)
    assert ex.value.errno == errno.EINVAL


def _encode_gop_parallel(enc, frames, streams=1, device=False):
    torch = None
    keep = []
    if device:
        import torch
    depth, out, q = enc.pipeline_depth, [], 0
    def take():
        enc.collect()
        out.append(enc.output() if streams == 1 else [enc.output(s) for s in range(streams)])
    for fr in frames:
        if q == depth:
            take()
            q -= 1
        for s in range(streams):
            y, uv = fr if streams == 1 else fr[s]
            if device:
                ty, tuv = torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda()
                keep.append((ty, tuv))
                torch.cuda.synchronize()
                enc.set_device_input(s, ty.data_ptr(), tuv.data_ptr())
            else:
                enc.set_input(y, uv, s)
        enc.submit()
        q += 1
    enc.flush()
    while q:
        take()
        q -= 1
    return out


@pytest.mark.parametrize("w,h,K,G,n,device", [(320, 240, 5, 4, 43, False), (320, 240, 5, 4, 20, True),
                                              (640, 368, 3, 8, 50, False), (64, 48, 1, 3, 7, False),
                                              (1920, 1080, 4, 3, 14, True)])
def test_gop_parallel_session_equals_the_serial_stream(ce, w, h, K, G, n, device):
    """gop_parallel = G: G consecutive closed GOPs of one stream run as lock-step lanes of the same
    launches (they depend on nothing of each other). Frames go in and come out in stream order; the
    bytes are the serial encode's, whole groups, partial last group and partial last GOP included."""
    frames = [ol.gen_frame("B", w, h, f) for f in range(n)]
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=27, keyint=K))
    want = [oe.encode(y, uv) for y, uv in frames]
    for low in (0, 1):   # with and without the second (G-1) K frames of look-ahead: the same bytes
        with ce.CedarEncoder(width=w, height=h, qp=27, keyframe_interval=K, gop_parallel=G, gop_low_latency=low) as enc:
            assert enc.pipeline_depth >= (2 - low) * (G - 1) * K + 2
            got = _encode_gop_parallel(enc, frames, device=device)
        bad = [i for i in range(n) if got[i] != want[i]]
        assert not bad, "first differing frames: %s" % bad[:8]


@pytest.mark.parametrize("at", [1, 3, 6])
def test_gop_parallel_failed_step_drops_one_frame_per_lane_and_nothing_drifts(ce, monkeypatch, at):
    """A lock-step step that fails takes one frame of every GOP lane with it (each reported by its own
    collect); the steps behind it are coded again on the state from before it, and every lane starts
    its next GOP on time. What is delivered must decode, picture for picture, to something close to
    its source: a reference that drifted would show at once."""
    from tests.indep import h264_refdec as rd
    w, h, K, G, n = 320, 240, 4, 3, 30
    frames = [ol.gen_frame("B", w, h, f) for f in range(n)]
    monkeypatch.setenv("CEDAR_FAULT_INJECT", "1")
    monkeypatch.setenv("CEDAR_FAULT_INJECT_AT", str(at))     # the at-th lock-step step
    got, failed = {}, []
    with ce.CedarEncoder(width=w, height=h, qp=26, keyframe_interval=K, gop_parallel=G) as enc:
        depth, q, nxt = enc.pipeline_depth, 0, 0

        def take():
            nonlocal nxt
            try:
                enc.collect()
                got[nxt] = enc.output()
            except ce.EncoderError as ex:
                assert ex.errno == errno.EIO
                failed.append(nxt)
            nxt += 1
        for f in range(n):
            if q == depth:
                take()
                q -= 1
            enc.set_input(*frames[f])
            enc.submit()
            q += 1
        enc.flush()
        while q:
            take()
            q -= 1
    grp, t = (at - 1) // K, (at - 1) % K              # the step's group of G GOPs and its position inside them
    assert failed == [(grp * G + g) * K + t for g in range(G)]
    dec = rd.Decoder()
    for f in sorted(got):
        pic, = dec.decode(got[f])
        y = np.frombuffer(pic.luma_bytes(), np.uint8).reshape(240, 320).astype(np.float64)
        mse = ((y - frames[f][0].reshape(h, w)) ** 2).mean()
        assert mse < 40, (f, mse)                      # ~35 dB at QP 26; a wrong reference gives hundreds
    kinds = {f: [got[f][i + 4] & 0x1f for i in range(len(got[f]) - 4) if got[f][i:i + 4] == b"\0\0\0\1"] for f in got}
    for f in got:                                      # every GOP starts with an IDR picture: its first delivered frame
        first_of_gop = min(x for x in got if x // K == f // K)
        assert (5 in kinds[f]) == (f == first_of_gop), (f, kinds[f])


def test_gop_parallel_two_streams_and_collect_before_input_is_there(ce):
    w, h, K, G, n = 320, 240, 4, 3, 19
    frames = [[ol.gen_frame("B", w, h, f, s << 8) for s in range(2)] for f in range(n)]
    want = []
    for s in range(2):
        oe = ol.OracleEncoder(ol.default_config(w, h, qp=30, keyint=K))
        want.append([oe.encode(*frames[f][s]) for f in range(n)])
    with ce.CedarEncoder(width=w, height=h, qp=30, keyframe_interval=K, gop_parallel=G, n_streams=2) as enc:
        for s in range(2):
            enc.set_input(*frames[0][s], stream=s)
        enc.submit()
        with pytest.raises(ce.EncoderError) as ex:   # frame 0 needs frame (G-1)*K too, or a flush
            enc.collect()
        assert ex.value.errno == errno.EAGAIN
        with pytest.raises(ce.EncoderError):          # lanes are fed through submit / collect / flush
            enc.encode()
    with ce.CedarEncoder(width=w, height=h, qp=30, keyframe_interval=K, gop_parallel=G, n_streams=2) as enc:
        got = _encode_gop_parallel(enc, frames, streams=2)
    for s in range(2):
        assert [g[s] for g in got] == want[s], s


@pytest.mark.parametrize("w,h,qp,n,keyint,slices,subpel,streams,frames", [
    (96, 64, 28, 3, 12, 1, 0, 1, 8), (320, 240, 30, 5, 25, 3, 2, 1, 8), (16, 16, 24, 3, 6, 1, 0, 1, 8),
    (70, 50, 20, 4, 9, 2, 1, 2, 7), (640, 368, 24, 11, 25, 1, 2, 1, 5), (1920, 1080, 26, 30, 25, 1, 0, 1, 3),
    (32, 160, 33, 3, 30, 2, 0, 3, 6)])
def test_cyclic_intra_refresh_equals_the_oracle(ce, w, h, qp, n, keyint, slices, subpel, streams, frames):
    """intra_refresh = N: in the t-th P picture the macroblocks with (x + 2*y + t) % N == 0 are coded
    Intra16x16 from the constant prediction (constrained_intra_pred in the PPS): k_inter's refresh branch,
    intra macroblock types and intra-aware vector prediction in both CAVLC kernels, bS 3/4 beside inter
    edges in the deblock kernel. Records, levels, reconstruction and bytes equal the scalar twin's,
    frames kept in flight; the independent decoder rebuilds the same pictures."""
    from tests.indep import h264_refdec as rd
    oes = [ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=keyint, slice_count=slices, subpel=subpel, intra_refresh=n))
           for _ in range(streams)]
    dec = rd.Decoder() if w <= 320 else None
    src = [[ol.gen_frame("BC"[s & 1], w, h, f, s << 8) for s in range(streams)] for f in range(frames)]
    with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint, slice_count=slices, subpel=subpel,
                         n_streams=streams, intra_refresh=n) as ge:
        for f in range(frames):   # one at a time: the whole state is compared
            for s in range(streams):
                ge.set_input(src[f][s][0], src[f][s][1], stream=s)
            ge.encode()
            refs = [oes[s].encode(*src[f][s]) for s in range(streams)]
            if streams == 1:
                _compare_state(oes[0], ge)
                mb = ge.debug_read("mbinfo")
                my, mx = np.mgrid[0:ge.mb_h, 0:ge.mb_w]
                t = f % keyint
                if t:
                    assert np.array_equal((mb["type"] <= 1).reshape(ge.mb_h, ge.mb_w), (mx + 2 * my + t) % n == 0)
            for s in range(streams):
                assert ge.output(s) == refs[s], "frame %d stream %d: bytestream differs" % (f, s)
            if dec:
                pic, = dec.decode(ge.output(0))
                assert pic.luma_bytes() == ge.debug_read("recon_y").tobytes()
    # the same stream with frames in flight
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=keyint, slice_count=slices, subpel=subpel, intra_refresh=n))
    want = [oe.encode(*src[f][0]) for f in range(frames)]
    with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint, slice_count=slices, subpel=subpel, intra_refresh=n) as ge:
        got, q = [], 0
        for f in range(frames):
            if q == ge.pipeline_depth:
                ge.collect(); got.append(ge.output(0)); q -= 1
            ge.set_input(src[f][0][0], src[f][0][1])
            ge.submit(); q += 1
        while q:
            ge.collect(); got.append(ge.output(0)); q -= 1
    assert got == want
    for bad in (1, 2, -3, 70000):
        with pytest.raises(ce.EncoderError):
            ce.CedarEncoder(width=w, height=h, intra_refresh=bad)


@pytest.mark.parametrize("w,h,qp,roi,keyint,slices,subpel,streams,frames,extra", [
    (96, 64, 30, [(1, 1, 3, 2, -8)], 12, 1, 0, 1, 6, {}),
    (320, 240, 28, [(0, 0, 4, 3, -6), (2, 1, 6, 4, 10), (19, 14, 1, 1, -20)], 25, 3, 2, 1, 6, {}),
    (112, 80, 24, [(3, 0, 2, 5, -12), (0, 2, 7, 1, 12)], 4, 1, 1, 2, 7, dict(intra_refresh=3)),
    (48, 48, 47, [(1, 1, 1, 1, -24)], 30, 1, 0, 1, 5, {}), (48, 48, 5, [(0, 0, 3, 3, -24)], 30, 1, 0, 1, 4, {}),
    (640, 368, 26, [(10, 5, 20, 12, -10), (0, 0, 40, 23, 6)], 25, 1, 2, 1, 4, {}),
    (1920, 1080, 30, [(40, 20, 40, 30, -8), (0, 60, 120, 8, 12)], 25, 1, 0, 1, 3, {}),
    (160, 96, 30, [(2, 1, 5, 3, -9)], 5, 2, 0, 3, 8, dict(gop_parallel=2)),
    # steps of 48 between neighbours and of 36 against the slice QP: mb_qp_delta goes round modulo 52 (7.4.5)
    (64, 48, 26, [(0, 0, 1, 1, 24), (1, 0, 1, 1, -24), (2, 1, 1, 2, 24)], 3, 1, 0, 1, 5, {}),
    (64, 48, 14, [(0, 0, 2, 1, 24)], 30, 2, 0, 1, 4, dict(mad_th=[0, 0, 0, 0], mad_dqp=[0, 0, 0, 0, 12])),
    # MAD classes: the QP follows how well the chosen prediction fits (on their own, with regions, with refresh)
    (96, 64, 30, [], 12, 1, 0, 1, 6, dict(mad_th=[2, 5, 10, 20], mad_dqp=[-4, -2, 0, 2, 4])),
    (320, 240, 28, [(2, 1, 6, 4, 10)], 25, 2, 2, 2, 6, dict(mad_th=[1, 3, 6, 12], mad_dqp=[6, 3, 0, -3, -6])),
    (112, 80, 24, [], 6, 1, 1, 1, 8, dict(mad_th=[4, 8, 16, 32], mad_dqp=[-12, 0, 0, 0, 12], intra_refresh=3)),
    (640, 368, 26, [], 25, 1, 0, 1, 4, dict(mad_th=[1, 2, 4, 8], mad_dqp=[-3, -1, 0, 2, 5])),
    (64, 48, 47, [], 8, 1, 0, 1, 4, dict(mad_th=[0, 0, 0, 0], mad_dqp=[0, 0, 0, 0, 5]))])
def test_regions_of_interest_equal_the_oracle(ce, w, h, qp, roi, keyint, slices, subpel, streams, frames, extra):
    """roi: macroblocks of a region are quantised at the picture's QP + the region's offset. Every kernel that
    derives something from the QP looks it up per macroblock (k_intra_decide, k_intra_recon2<ROI>, k_inter),
    k_qp_eff scans QP_Y as the decoder will hold it (a macroblock without coded blocks inherits) and lays down
    the deblocking thresholds per edge, deblock mode 3 filters with them, CAVLC writes mb_qp_delta.
    Records, levels, reconstruction and bytes equal the scalar twin's; the independent decoder agrees."""
    from tests.indep import h264_refdec as rd
    G = extra.get("gop_parallel", 0)
    okw = {k: v for k, v in extra.items() if k != "gop_parallel"}
    src = [[ol.gen_frame("BC"[s & 1], w, h, f, s << 8) for s in range(streams)] for f in range(frames)]
    oes = [ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=keyint, slice_count=slices, subpel=subpel, roi=roi, **okw))
           for _ in range(streams)]
    want = [[oes[s].encode(*src[f][s]) for f in range(frames)] for s in range(streams)]
    if G:
        with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint, slice_count=slices, subpel=subpel,
                             n_streams=streams, roi=tuple(roi), **extra) as ge:
            got = _encode_gop_parallel(ge, [[src[f][s] for s in range(streams)] for f in range(frames)], streams=streams)
        for s in range(streams):
            assert [g[s] for g in got] == want[s], s
        return
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=keyint, slice_count=slices, subpel=subpel, roi=roi, **okw))
    dec = rd.Decoder() if w <= 320 else None
    with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint, slice_count=slices, subpel=subpel,
                         n_streams=streams, roi=tuple(roi), **extra) as ge:
        for f in range(frames):
            for s in range(streams):
                ge.set_input(src[f][s][0], src[f][s][1], stream=s)
            ge.encode()
            oe.encode(*src[f][0])
            _compare_state(oe, ge)   # stream 0: records, levels, reconstruction
            assert np.array_equal(oe.mb_qp(), ge.debug_read("mb_qp")), f   # QP_Y as the decoder will hold it
            for s in range(streams):
                assert ge.output(s) == want[s][f], "frame %d stream %d: bytestream differs" % (f, s)
            if dec:
                pic, = dec.decode(ge.output(0))
                assert pic.luma_bytes() == ge.debug_read("recon_y").tobytes()
    # frames in flight
    with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint, slice_count=slices, subpel=subpel,
                         roi=tuple(roi), **extra) as ge:
        got, q = [], 0
        for f in range(frames):
            if q == min(ge.pipeline_depth, 3):
                ge.collect(); got.append(ge.output(0)); q -= 1
            ge.set_input(src[f][0][0], src[f][0][1])
            ge.submit(); q += 1
        while q:
            ge.collect(); got.append(ge.output(0)); q -= 1
    assert got == want[0]
    for bad in ([(0, 0, 0, 1, 3)], [(-1, 0, 1, 1, 3)], [(0, 0, 1, 1, 25)], [(0, 0, 1, 1, 1)] * 5):
        with pytest.raises(ce.EncoderError):
            ce.CedarEncoder(width=w, height=h, roi=tuple(bad))
    for bad in (dict(mad_th=(5, 4, 6, 7), mad_dqp=(1, 0, 0, 0, 0)), dict(mad_dqp=(13, 0, 0, 0, 0)), dict(mad_th=(0, 0, 0, 256))):
        with pytest.raises(ce.EncoderError):
            ce.CedarEncoder(width=w, height=h, **bad)


@pytest.mark.parametrize("mode", ["plain", "narrow", "wide", "r2", "refresh", "roi", "roi-pairs", "wide-rows", "narrow-rows"])
def test_fuzz_sweep_against_oracle(mode):
    """tools/fuzz_parity.py with a fixed seed: random geometry (every other picture 1-3 macroblocks
    wide or high in "narrow"), QP, GOP length, slices, streams (up to 33 in "wide"), search range,
    chroma QP offset, header options and frames in flight; every access unit against the oracle.
    The sweep is what found k_cavlc2 reading record -1 on pictures one macroblock wide."""
    import subprocess
    import sys
    env = dict(os.environ)
    # the two deblock kernels: the session picks one by its unit count (enc_plan.h); "-pairs" runs everything on the wave
    # pairs of k_deblock2_planes, "-rows" everything on k_deblock_rows, lock-step streams and GOP lanes included
    n_cases = 14
    if mode.endswith("-pairs") or mode.endswith("-rows"):
        env["CEDAR_DEBLOCK_ROWS"] = "0" if mode.endswith("-pairs") else "2"
        mode = mode.rsplit("-", 1)[0]
        n_cases = 9
    if mode == "narrow":
        env["FUZZ_NARROW"] = "1"
    if mode == "wide":
        env["FUZZ_WIDE"] = "1"
    if mode == "r2":   # round 2's options on top: sub-sample vectors, the macroblock bit cap, rate control, GOP lanes
        env["FUZZ_R2"] = "1"
    if mode == "refresh":   # ... and cyclic intra refresh on top of those
        env["FUZZ_R2"] = env["FUZZ_REFRESH"] = "1"
    if mode == "roi":       # ... and regions of interest on top of all of it
        env["FUZZ_R2"] = env["FUZZ_REFRESH"] = env["FUZZ_ROI"] = env["FUZZ_MAD"] = "1"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "20261003", str(n_cases)],
                       capture_output=True, text=True, env=env, timeout=300, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "%d cases bit-exact" % n_cases in r.stdout
