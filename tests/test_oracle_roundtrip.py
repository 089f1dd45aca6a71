"""Closed-loop conformance of the oracle: decode(stream) == in-loop recon.

The container has no H.264 decoder, so the stream is decoded
by the independently written test decoder (oracle/h264o_dec.c) and compared
sample for sample with the encoder's deblocked reconstruction.
"""
import numpy as np
import pytest

from tests import oracle_lib as ol

CASES = [
    # (w, h, qp, keyint, frames, pattern, slices)
    (64, 48, 30, 1, 3, "A", 1),
    (64, 48, 30, 1, 3, "B", 1),
    (64, 48, 24, 1, 2, "C", 1),
    (320, 240, 30, 1, 3, "B", 1),   # BASELINE configs[0] shape (I only)
    (64, 48, 30, 25, 6, "A", 1),
    (64, 48, 30, 25, 6, "B", 1),
    (320, 240, 30, 25, 5, "B", 1),
    (64, 48, 24, 25, 4, "C", 1),
    (70, 50, 24, 25, 4, "B", 1),    # not a multiple of 16: padding + cropping
    (320, 240, 24, 25, 4, "B", 4),  # config[3] structure: 4 slices, idc 2
    (48, 80, 24, 4, 6, "B", 5),     # ragged slice split 1/1/1/1/1
    (64, 48, 5, 3, 4, "C", 1),      # very low QP: big levels, escape codes
    (64, 48, 47, 3, 4, "B", 1),     # maximum QP the reference accepts
    (16, 16, 24, 2, 4, "C", 1),     # one macroblock
    (432, 32, 24, 2, 3, "B", 1),    # wide and flat
]


@pytest.mark.parametrize("w,h,qp,keyint,frames,pattern,slices", CASES)
def test_decode_equals_recon(w, h, qp, keyint, frames, pattern, slices):
    cfg = ol.default_config(w, h, qp=qp, keyint=keyint, slice_count=slices)
    enc, dec = ol.OracleEncoder(cfg), ol.OracleDecoder()
    for f in range(frames):
        y, uv = ol.gen_frame(pattern, w, h, f)
        au = enc.encode(y, uv)
        assert dec.decode(au) == 1
        ry, ruv = enc.recon()
        dy, duv = dec.picture()
        assert np.array_equal(ry, dy), "luma mismatch frame %d" % f
        assert np.array_equal(ruv, duv), "chroma mismatch frame %d" % f
        assert dec.crop_size() == (w, h)
        # the reconstruction must resemble the source
        src = y.reshape(h, w).astype(np.float64)
        mse = ((src - ry[:h, :w]) ** 2).mean()
        assert mse < (400 if qp <= 30 else 3000)


def test_static_scene_is_all_skip():
    """pattern A is static: every P picture collapses to one mb_skip_run."""
    cfg = ol.default_config(320, 240, qp=30, keyint=25)
    enc = ol.OracleEncoder(cfg)
    y, uv = ol.gen_frame("A", 320, 240, 0)
    enc.encode(y, uv)
    au = enc.encode(y, uv)
    assert len(au) < 16
    assert all(m.type == ol.MB_PSKIP for m in enc.mbs())


def test_global_motion_is_found():
    """pattern B moves by (3,2) px/frame: content at x in frame f-1 sits at x-3
    in frame f, so the best match for a block is at (+3,+2) in the reference."""
    cfg = ol.default_config(320, 240, qp=24, keyint=25)
    enc = ol.OracleEncoder(cfg)
    for f in range(2):
        enc.encode(*ol.gen_frame("B", 320, 240, f))
    mb = enc.mb_array()
    inner = mb.reshape(15, 20)[2:-2, 2:-2]
    frac = ((inner["mvx"] == 12) & (inner["mvy"] == 8)).mean()   # vectors are kept in quarter samples
    assert frac > 0.8


def test_no_start_code_emulation_in_payload():
    cfg = ol.default_config(64, 48, qp=5, keyint=1)
    enc = ol.OracleEncoder(cfg)
    au = enc.encode(*ol.gen_frame("A", 64, 48, 0))
    # inside any NAL payload no 00 00 00/01/02 may appear
    starts = [i for i in range(len(au) - 3) if au[i:i + 4] == b"\0\0\0\1"]
    for k, s in enumerate(starts):
        e = starts[k + 1] if k + 1 < len(starts) else len(au)
        payload = au[s + 4:e]
        for i in range(len(payload) - 2):
            assert not (payload[i] == 0 and payload[i + 1] == 0 and payload[i + 2] < 3)


@pytest.mark.parametrize("off", [-12, -5, 0, 4, 12])
def test_chroma_qp_offset_round_trip(off):
    """chroma_qp_index_offset over its whole range (the reference fixes +4, cedar.c:971-972)."""
    w, h = 96, 64
    enc = ol.OracleEncoder(ol.default_config(w, h, qp=30, keyint=3, chroma_qp_offset=off))
    dec = ol.OracleDecoder()
    sizes = []
    for f in range(5):
        au = enc.encode(*ol.gen_frame("B", w, h, f))
        sizes.append(len(au))
        assert dec.decode(au) == 1
        ry, ruv = enc.recon()
        dy, duv = dec.picture()
        assert np.array_equal(dy, ry) and np.array_equal(duv, ruv)
    for bad in (-13, 13):
        with pytest.raises(Exception):
            ol.OracleEncoder(ol.default_config(w, h, chroma_qp_offset=bad))


def test_fuzzed_round_trips():
    """tools/fuzz_oracle_roundtrip.py, fixed seed: 120 random configurations (geometry, QP, GOP,
    slices, search range, chroma QP offset, header options), decode(stream) == reconstruction."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_oracle_roundtrip.py"), "20261003", "120"],
                       capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-1000:])
    assert "120 cases, decode == recon" in r.stdout


def test_rate_control_reaches_the_target_and_stays_decodable():
    """cedar_enc_config_ext.bitrate / h264o_config.bitrate: one QP per picture (slice_qp_delta) from an
    integer law that only looks 17 and more frames back. After the start-up the stream runs within
    one QP step (12 %) below the target; every picture still decodes to the encoder's reconstruction,
    with the independent decoder too (it follows slice_qp_delta)."""
    import numpy as np
    from tests.indep import h264_refdec as rd
    w, h, n = 192, 112, 110
    for target in (150000, 600000):
        oe = ol.OracleEncoder(ol.default_config(w, h, qp=30, keyint=25, bitrate=target))
        dec, pdec = ol.OracleDecoder(), rd.Decoder()
        sizes, qps = [], []
        for f in range(n):
            au = oe.encode(*ol.gen_frame("B", w, h, f))
            sizes.append(len(au))
            qps.append(oe.frame_qp())
            assert dec.decode(au) == 1
            ry, ruv = oe.recon()
            dy, duv = dec.picture()
            assert np.array_equal(dy, ry) and np.array_equal(duv, ruv)
            if f % 9 == 0 or f < 3:
                pic = pdec.decode(au)[0]
                assert pic.luma_bytes() == ry.tobytes() and pic.nv12_chroma_bytes() == ruv.tobytes()
            else:
                pdec.decode(au)
        assert qps[:17] == [30] * 17 and len(set(qps)) > 1       # nothing is known before frame 17
        rate = sum(sizes[60:110]) * 8 * 30 / 50.0               # two whole GOPs
        assert 0.80 * target <= rate <= 1.02 * target, (target, rate, qps[::10])


def test_rate_control_by_groups_of_gop_lanes():
    """h264o_config.rc_lanes = G (the law of a gop_parallel session): one QP per group of G GOPs, the
    configured one for the first 1 + ceil(16 / keyint) groups, then what the group that far back says; streams stay
    decodable by both decoders, and rc_lanes 0 / 1 is the law by frames."""
    import numpy as np
    from tests.indep import h264_refdec as rd
    w, h, K, G, n = 192, 112, 4, 3, 132
    frames = [ol.gen_frame("BC"[(f // 7) % 2], w, h, f) for f in range(n)]
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=30, keyint=K, bitrate=1480000, rc_lanes=G))
    dec, pdec = ol.OracleDecoder(), rd.Decoder()
    qps = []
    for f, (y, uv) in enumerate(frames):
        au = oe.encode(y, uv)
        qps.append(oe.frame_qp())
        assert dec.decode(au) == 1
        ry, ruv = oe.recon()
        dy, duv = dec.picture()
        assert np.array_equal(dy, ry) and np.array_equal(duv, ruv)
        pic = pdec.decode(au)[0]
        if f % 11 == 0:
            assert pic.luma_bytes() == ry.tobytes()
    lag = 1 + (16 + K - 1) // K
    groups = [qps[i:i + G * K] for i in range(0, n, G * K)]
    assert all(len(set(g)) == 1 for g in groups)
    assert [g[0] for g in groups[:lag]] == [30] * lag and len({g[0] for g in groups}) > 2
    by_frames = ol.OracleEncoder(ol.default_config(w, h, qp=30, keyint=K, bitrate=1480000))
    one_lane = ol.OracleEncoder(ol.default_config(w, h, qp=30, keyint=K, bitrate=1480000, rc_lanes=1))
    assert all(by_frames.encode(y, uv) == one_lane.encode(y, uv) for y, uv in frames[:40])
    with pytest.raises(Exception):
        ol.OracleEncoder(ol.default_config(w, h, qp=30, keyint=K, bitrate=1480000, rc_lanes=2, scene_cut=30))


@pytest.mark.parametrize("w,h,n,slices,subpel,pattern,keyint", [
    (96, 64, 3, 1, 0, "B", 12), (112, 80, 5, 2, 2, "B", 25), (64, 48, 4, 1, 0, "A", 8),
    (48, 48, 7, 1, 1, "C", 30), (16, 16, 3, 1, 0, "C", 6), (160, 32, 3, 2, 2, "B", 9)])
def test_cyclic_intra_refresh(w, h, n, slices, subpel, pattern, keyint):
    """h264o_config.intra_refresh = N: in the t-th P picture the macroblocks with (x + 2*y + t) % N == 0 are
    Intra16x16 (the PPS says constrained_intra_pred), N pictures in a row renew every macroblock, and
    both decoders -- the C one and the independent Python one -- rebuild the encoder's pictures."""
    from tests.indep import h264_refdec as rd
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=28, keyint=keyint, slice_count=slices, subpel=subpel, intra_refresh=n))
    dec, pdec = ol.OracleDecoder(), rd.Decoder()
    mb_w, mb_h = (w + 15) // 16, (h + 15) // 16
    my, mx = np.mgrid[0:mb_h, 0:mb_w]
    seen = np.zeros((mb_h, mb_w), int)
    for f in range(2 + n):
        au = oe.encode(*ol.gen_frame(pattern, w, h, f))
        assert dec.decode(au) == 1, dec.error()
        ry, ruv = oe.recon()
        dy, duv = dec.picture()
        assert np.array_equal(ry, dy) and np.array_equal(ruv, duv), f
        pic, = pdec.decode(au)
        assert pic.luma_bytes() == ry.tobytes() and pic.nv12_chroma_bytes() == ruv.tobytes(), f
        intra = (oe.mb_array()["type"] <= ol.MB_I16x16).reshape(mb_h, mb_w)
        t = f % keyint
        if t:
            assert np.array_equal(intra, (mx + 2 * my + t) % n == 0), f
            seen += intra
    assert seen.min() >= 1
    for bad in (1, 2, -1, 70000):
        with pytest.raises(Exception):
            ol.OracleEncoder(ol.default_config(w, h, intra_refresh=bad))


@pytest.mark.parametrize("w,h,qp,roi,slices,subpel,pattern,keyint,extra", [
    (96, 64, 30, [(1, 1, 3, 2, -8)], 1, 0, "B", 12, {}),
    (160, 96, 28, [(0, 0, 4, 3, -6), (2, 1, 6, 4, 10), (9, 5, 1, 1, -20)], 2, 2, "B", 25, {}),
    (112, 80, 24, [(3, 0, 2, 5, -12), (0, 2, 7, 1, 12)], 1, 1, "C", 4, dict(intra_refresh=3)),
    (48, 48, 47, [(1, 1, 1, 1, -24)], 1, 0, "B", 30, {}), (48, 48, 5, [(0, 0, 3, 3, -24)], 1, 0, "C", 30, {}),
    # a step of 48 between neighbours, and of 36 against the slice QP at the slice's first macroblock: mb_qp_delta
    # only reaches -26..+25 (7.4.5) and goes round modulo 52
    (64, 48, 26, [(0, 0, 1, 1, 24), (1, 0, 1, 1, -24), (2, 1, 1, 2, 24)], 1, 0, "C", 3, {}),
    (64, 48, 14, [(0, 0, 2, 1, 24)], 2, 0, "C", 30, dict(mad_th=[0, 0, 0, 0], mad_dqp=[0, 0, 0, 0, 12]))])
def test_regions_of_interest(w, h, qp, roi, slices, subpel, pattern, keyint, extra):
    """h264o_config.roi: the macroblocks of a region are quantised at slice QP + offset (first region wins,
    1..51), carried by mb_qp_delta; a macroblock without one (no coded block, P_Skip) keeps its
    predecessor's QP_Y, which is what deblocking then uses for it. Both decoders rebuild the pictures."""
    from tests.indep import h264_refdec as rd
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=keyint, slice_count=slices, subpel=subpel, roi=roi, **extra))
    dec, pdec = ol.OracleDecoder(), rd.Decoder()
    mb_w, mb_h = (w + 15) // 16, (h + 15) // 16
    want = np.full((mb_h, mb_w), qp)
    for x, y, rw, rh, dq in reversed(roi):
        want[y:y + rh, x:x + rw] = min(51, max(1, qp + dq))
    for f in range(6):
        au = oe.encode(*ol.gen_frame(pattern, w, h, f))
        assert dec.decode(au) == 1, dec.error()
        ry, ruv = oe.recon()
        dy, duv = dec.picture()
        assert np.array_equal(ry, dy) and np.array_equal(ruv, duv), f
        pic, = pdec.decode(au)
        assert pic.luma_bytes() == ry.tobytes() and pic.nv12_chroma_bytes() == ruv.tobytes(), f
        mb = oe.mb_array()
        coded = ((mb["type"] == ol.MB_I16x16) | ((mb["cbp"] & 0x3f) != 0)).reshape(mb_h, mb_w)
        got = oe.mb_qp().reshape(mb_h, mb_w)
        if "mad_dqp" not in extra:
            assert np.array_equal(got[coded], want[coded]), f   # a coded macroblock has the QP of its region
    assert pdec.cover["mb_qp_delta_nonzero"] > 0
    for bad in ([(0, 0, 0, 1, 3)], [(-1, 0, 1, 1, 3)], [(0, 0, 1, 1, 25)], [(0, 0, 1, 1, 1)] * 5):
        with pytest.raises(Exception):
            ol.OracleEncoder(ol.default_config(w, h, roi=bad))


@pytest.mark.parametrize("w,h,qp,keyint,pattern,kw,want_qps", [
    (96, 64, 30, 12, "B", dict(mad_th=[2, 5, 10, 20], mad_dqp=[-4, -2, 0, 2, 4]), None),
    (160, 96, 28, 25, "B", dict(mad_th=[1, 3, 6, 12], mad_dqp=[6, 3, 0, -3, -6], roi=[(2, 1, 6, 4, 10)], subpel=2, slice_count=2), None),
    (112, 80, 24, 6, "C", dict(mad_th=[4, 8, 16, 32], mad_dqp=[-12, 0, 0, 0, 12], intra_refresh=3), {36}),   # noise: the last class everywhere
    (64, 48, 47, 8, "A", dict(mad_th=[0, 0, 0, 0], mad_dqp=[0, 0, 0, 0, 5]), {51})])                         # clipped at 51
def test_mad_classes(w, h, qp, keyint, pattern, kw, want_qps):
    """h264o_config.mad_th / mad_dqp: the macroblock's luma SAD / 256 against the prediction it chose picks one of
    five QP offsets (on top of the picture's QP and its region's); both decoders rebuild the pictures."""
    from tests.indep import h264_refdec as rd
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=keyint, **kw))
    dec, pdec = ol.OracleDecoder(), rd.Decoder()
    qs = set()
    for f in range(7):
        au = oe.encode(*ol.gen_frame(pattern, w, h, f))
        assert dec.decode(au) == 1, dec.error()
        ry, ruv = oe.recon()
        dy, duv = dec.picture()
        assert np.array_equal(ry, dy) and np.array_equal(ruv, duv), f
        pic, = pdec.decode(au)
        assert pic.luma_bytes() == ry.tobytes() and pic.nv12_chroma_bytes() == ruv.tobytes(), f
        qs |= set(oe.mb_qp().tolist())
    assert len(qs) > 1 or want_qps
    if want_qps:
        assert qs == want_qps
    for bad in (dict(mad_th=[5, 4, 6, 7]), dict(mad_dqp=[13, 0, 0, 0, 0]), dict(mad_th=[0, 0, 0, 256])):
        with pytest.raises(Exception):
            ol.OracleEncoder(ol.default_config(w, h, **bad))


def test_fuzzed_round_trips_with_round_2_options():
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_oracle_roundtrip.py"), "20261004", "80"],
                       capture_output=True, text=True, timeout=300, cwd=root, env=dict(os.environ, FUZZ_R2="1"))
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-1000:])
    assert "80 cases, decode == recon" in r.stdout


@pytest.mark.parametrize("w,h,qp,keyint,patterns,frames,kw,want_pcm", [
    (96, 64, 1, 3, "C", 4, {}, "all"), (96, 64, 10, 3, "C", 4, {}, "all"), (96, 64, 16, 3, "C", 4, {}, "all"),
    (96, 64, 18, 3, "C", 4, {}, "some"), (96, 64, 22, 3, "C", 4, {}, "none"), (96, 64, 1, 3, "B", 4, {}, "none"),
    (96, 80, 12, 4, "CB", 6, dict(slice_count=2), "all"), (64, 48, 8, 25, "C", 3, dict(subpel=2), "all"),
    (16, 48, 6, 2, "C", 3, {}, "all"), (64, 48, 5, 3, "C", 3, dict(pcm=-1), "none"),
    (64, 48, 5, 3, "C", 3, dict(intra_refresh=3), "none"), (64, 48, 5, 3, "C", 3, dict(roi=[(0, 0, 1, 1, 2)]), "none")])
def test_macroblocks_that_could_pass_3200_bits_are_coded_i_pcm(w, h, qp, keyint, patterns, frames, kw, want_pcm):
    """h264o_config.pcm (default on): a macroblock whose CAVLC bits could pass the 3200 of H.264 A.3.1 is coded
    I_PCM -- in I pictures from the open loop's levels with a tenth of room, in P pictures from the final levels.
    Both decoders rebuild the pictures (the independent one first showed that an I_PCM macroblock leaves QP_Y as it
    finds it, 7.4.5: only deblocking takes it as 0), no macroblock passes the limit, and the sessions that do not
    take I_PCM (regions, refresh, pcm = -1) are as before."""
    from tests.indep import h264_refdec as rd
    cfg = dict(pcm=0)
    cfg.update(kw)
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=keyint, **cfg))
    dec, pdec = ol.OracleDecoder(), rd.Decoder()
    n_pcm = n_mb = 0
    for f in range(frames):
        au = oe.encode(*ol.gen_frame(patterns[f % len(patterns)], w, h, f))
        assert dec.decode(au) == 1
        ry, ruv = oe.recon()
        dy, duv = dec.picture()
        assert np.array_equal(ry, dy) and np.array_equal(ruv, duv), f
        pic, = pdec.decode(au)
        assert pic.luma_bytes() == ry.tobytes() and pic.nv12_chroma_bytes() == ruv.tobytes(), f
        mb = oe.mb_array()
        pcm = mb["type"] == ol.MB_IPCM
        n_pcm += int(pcm.sum())
        n_mb += pcm.size
        if pcm.any():   # its reconstruction is its source, its neighbours count 16 coefficients in every block
            y, uv = ol.gen_frame(patterns[f % len(patterns)], w, h, f)
            src = np.zeros_like(ry)
            src[:h, :w] = y.reshape(h, w)
            src[:h, w:] = src[:h, w - 1:w]
            src[h:, :] = src[h - 1:h, :]
            for i in np.flatnonzero(pcm):
                mx, my = i % ((w + 15) // 16), i // ((w + 15) // 16)
                assert np.array_equal(ry[16 * my:16 * my + 16, 16 * mx:16 * mx + 16], src[16 * my:16 * my + 16, 16 * mx:16 * mx + 16])
            assert (mb["nnz"][pcm] == 16).all()
        if want_pcm != "none" or "pcm" not in kw and not kw.get("roi") and not kw.get("intra_refresh"):
            assert oe.max_mb_bits() <= 3200, (f, oe.max_mb_bits())
    assert {"all": n_pcm == n_mb, "some": 0 < n_pcm < n_mb, "none": n_pcm == 0}[want_pcm], (n_pcm, n_mb)
    if want_pcm != "none":
        assert pdec.cover["mb_I_PCM"] == n_pcm and not pdec.level_violations()


@pytest.mark.parametrize("w,h,qp,before,after,cut_at,kw", [
    (320, 240, 30, "B", "A", 5, {}), (320, 240, 30, "A", "B", 5, {}), (320, 240, 28, "B", "C", 3, dict(subpel=2)),
    (320, 240, 30, "A", "B", 1, dict(slice_count=2)), (160, 96, 24, "B", "A", 7, dict(keyint=8))])
def test_scene_cuts_become_idr_pictures(w, h, qp, before, after, cut_at, kw):
    """h264o_config.scene_cut = 50: a P picture half of whose macroblocks are predicted better by their own mean than
    by the best vector is coded as an IDR picture that starts a new GOP (two IDR pictures in a row still differ in
    idr_pic_id). At the cut the access unit is smaller and no worse than the P picture it replaces; away from cuts
    nothing changes; both decoders rebuild every picture."""
    from tests.indep import h264_refdec as rd
    n, keyint = cut_at + 5, kw.pop("keyint", 25)

    def run(sc):
        oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=keyint, scene_cut=sc, **kw))
        dec, pdec = ol.OracleDecoder(), rd.Decoder()
        aus, idr, err = [], [], []
        for f in range(n):
            y, uv = ol.gen_frame(before if f < cut_at else after, w, h, f)
            au = oe.encode(y, uv)
            assert dec.decode(au) == 1
            ry, ruv = oe.recon()
            dy, duv = dec.picture()
            assert np.array_equal(ry, dy) and np.array_equal(ruv, duv), f
            pic, = pdec.decode(au)
            assert pic.luma_bytes() == ry.tobytes() and pic.nv12_chroma_bytes() == ruv.tobytes(), f
            aus.append(au)
            idr.append(ol.lib().h264o_last_frame_type(oe.h))
            err.append(float(np.mean((ry[:h, :w].astype(np.int64) - y.reshape(h, w)) ** 2)))
        return aus, idr, err

    plain, idr0, err0 = run(0)
    cut, idr1, err1 = run(50)
    assert plain[:cut_at] == cut[:cut_at]                     # nothing changes before the cut
    assert idr0[cut_at] == 0 and idr1[cut_at] == 1
    if after == "C":   # noise: the intra picture costs about the same bytes and is a good deal better (finer dead zone)
        assert len(cut[cut_at]) < 1.05 * len(plain[cut_at]) and err1[cut_at] < 0.8 * err0[cut_at]
    else:
        assert len(cut[cut_at]) < len(plain[cut_at]) and err1[cut_at] <= err0[cut_at] * 1.05
    if after != "C":                                          # (noise: every picture is a cut)
        want = [int(f == 0 or f == cut_at or (f > cut_at and (f - cut_at) % keyint == 0) or (f < cut_at and f % keyint == 0)) for f in range(n)]
        assert idr1 == want                                   # the GOP restarts at the cut
    for bad in (-1, 101):
        with pytest.raises(Exception):
            ol.OracleEncoder(ol.default_config(w, h, scene_cut=bad))
