"""The reference's command line (userspace/h264enc.c:134-201) on both twins:
  * oracle/build/h264enc_scalar -- BASELINE configs[0]: 320x240 nv12, I-frame only,
    QP 30, scalar C macroblock loop, no GPU, no /dev/cedar (runs on CPU here)
  * cedarx_h264_encoder_amd/lib/h264enc -- the HIP path (GPU only)
Same argv, stdin pumping, short-read behaviour and exit codes."""
import os
import subprocess

import numpy as np
import pytest

from tests import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCALAR = os.path.join(ROOT, "oracle", "build", "h264enc_scalar")
HIPCLI = os.path.join(ROOT, "cedarx_h264_encoder_amd", "lib", "h264enc")


def _raw(pattern, w, h, frames):
    out = bytearray()
    for f in range(frames):
        y, uv = ol.gen_frame(pattern, w, h, f)
        out += y.tobytes() + uv.tobytes()
    return bytes(out)


def _decode_all(stream, frames):
    dec = ol.OracleDecoder()
    starts = [i for i in range(len(stream) - 4) if stream[i:i + 4] == b"\0\0\0\1"]
    # split into access units at each slice NAL (type 1 or 5) with first_mb == 0 is overkill here:
    # one slice per picture, so every slice NAL ends an access unit
    n, au_start = 0, 0
    for k, s in enumerate(starts):
        t = stream[s + 4] & 0x1f
        if t in (1, 5):
            e = starts[k + 1] if k + 1 < len(starts) else len(stream)
            assert dec.decode(stream[au_start:e]) == 1
            au_start = e
            n += 1
    assert n == frames
    return dec


def test_scalar_cli_config0_320x240_intra_only(tmp_path):
    """configs[0]: 320x240 nv12 I-frame-only QP=30 through the scalar C loop, via stdin."""
    w, h, frames = 320, 240, 4
    raw = _raw("B", w, h, frames)
    out = tmp_path / "out.264"
    env = dict(os.environ, H264ENC_QP="30", H264ENC_KEYINT="1")
    r = subprocess.run([SCALAR, "-", str(w), str(h), str(out)], input=raw + b"\x00" * 100,  # trailing partial frame is dropped
                       capture_output=True, env=env, timeout=120)
    assert r.returncode == 0
    assert b"Frame     3:" in r.stdout
    stream = out.read_bytes()
    enc = ol.OracleEncoder(ol.default_config(w, h, qp=30, keyint=1))
    want = b"".join(enc.encode(*ol.gen_frame("B", w, h, f)) for f in range(frames))
    assert stream == want
    dec = _decode_all(stream, frames)
    assert dec.crop_size() == (w, h)
    nal_types = [stream[i + 4] & 0x1f for i in range(len(stream) - 4) if stream[i:i + 4] == b"\0\0\0\1"]
    assert nal_types == [7, 8, 5, 5, 5, 5]


def test_scalar_cli_usage_and_errors(tmp_path):
    for env in (dict(H264ENC_MAD="1,2,3:4"), dict(H264ENC_ROI="1,1,2,2,-4;3,3,x"), dict(H264ENC_ROI="0,0,1,1,1;" * 5)):
        r = subprocess.run([SCALAR, "-", "64", "48", str(tmp_path / "o.264")], input=b"", capture_output=True, env=dict(os.environ, **env))
        assert r.returncode != 0 and (b"H264ENC_MAD wants" in r.stderr or b"H264ENC_ROI wants" in r.stderr), env
    r = subprocess.run([SCALAR, "-", "320"], capture_output=True)
    assert r.returncode == 255 and b"Usage:" in r.stdout      # return -1 (h264enc.c:141-144)
    r = subprocess.run([SCALAR, str(tmp_path / "missing.yuv"), "320", "240", str(tmp_path / "o.264")], capture_output=True)
    assert r.returncode == 255 and b"Failed to open input file" in r.stderr
    r = subprocess.run([SCALAR, "-", "321", "240", str(tmp_path / "o.264")], input=b"", capture_output=True)
    assert r.returncode != 0 and b"config failed" in r.stderr  # odd width: EINVAL (cedar.c:749)


def test_h264dec_tool_reports_psnr(tmp_path):
    """the stand-alone decoder tool decodes what the CLI wrote and reports PSNR."""
    w, h, frames = 320, 240, 3
    raw = _raw("B", w, h, frames)
    src, out, dec = tmp_path / "src.nv12", tmp_path / "o.264", tmp_path / "dec.nv12"
    src.write_bytes(raw)
    env = dict(os.environ, H264ENC_QP="30", H264ENC_KEYINT="25")
    assert subprocess.run([SCALAR, str(src), str(w), str(h), str(out)], capture_output=True, env=env).returncode == 0
    tool = os.path.join(ROOT, "oracle", "build", "h264dec")
    r = subprocess.run([tool, str(out), str(dec), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "decoded 3 pictures 320x240" in r.stdout
    psnr = float(r.stdout.strip().split("PSNR Y ")[-1].split(" dB")[0])
    assert 30.0 < psnr < 60.0
    assert dec.stat().st_size == len(raw)
    # decoded pictures equal the encoder's reconstruction (cropped)
    enc = ol.OracleEncoder(ol.default_config(w, h, qp=30, keyint=25))
    data = dec.read_bytes()
    for f in range(frames):
        enc.encode(*ol.gen_frame("B", w, h, f))
        ry, ruv = enc.recon()
        fr = np.frombuffer(data[f * w * h * 3 // 2:(f + 1) * w * h * 3 // 2], np.uint8)
        assert np.array_equal(fr[:w * h].reshape(h, w), ry[:h, :w])
        assert np.array_equal(fr[w * h:].reshape(h // 2, w), ruv[:h // 2, :w])
    # a corrupted stream is rejected, not silently accepted
    bad = bytearray(out.read_bytes())
    bad[len(bad) // 2] ^= 0xFF
    (tmp_path / "bad.264").write_bytes(bytes(bad))
    r = subprocess.run([tool, str(tmp_path / "bad.264"), "-"], capture_output=True, text=True)
    assert r.returncode in (0, 1)  # may still parse; must not crash


@pytest.mark.gpu
def test_hip_cli_config1_full_length_pipelined_and_serial(tmp_path):
    """BASELINE configs[1] as specified: 854x480, 50 frames (2 GOPs of 25), QP 30, through the
    reference command line; the pipelined frame pump and the serial one (h264enc.c:181-198)
    write the same file, and it equals the scalar C twin's."""
    w, h, frames = 854, 480, 50
    raw = _raw("B", w, h, frames)
    env = dict(os.environ, H264ENC_QP="30", H264ENC_KEYINT="25")
    outs = {}
    for mode in ("1", "0"):
        o = tmp_path / ("hip%s.264" % mode)
        r = subprocess.run([HIPCLI, "-", str(w), str(h), str(o)], input=raw, capture_output=True,
                           env=dict(env, H264ENC_PIPELINE=mode), timeout=300)
        assert r.returncode == 0, r.stderr
        assert b"Frame    49:" in r.stdout
        outs[mode] = o.read_bytes()
    assert outs["1"] == outs["0"]
    enc = ol.OracleEncoder(ol.default_config(w, h, qp=30, keyint=25))
    want = b"".join(enc.encode(*ol.gen_frame("B", w, h, f)) for f in range(frames))
    assert outs["1"] == want
    nal = [outs["1"][i + 4] & 0x1f for i in range(len(outs["1"]) - 4) if outs["1"][i:i + 4] == b"\0\0\0\1"]
    assert nal == [7, 8, 5] + [1] * 24 + [5] + [1] * 24


@pytest.mark.gpu
def test_hip_cli_equals_scalar_cli(tmp_path):
    """README.md:25-36 use: raw nv12 on stdin, 854x480 (not a multiple of 16), I+P."""
    w, h, frames = 854, 480, 5
    raw = _raw("B", w, h, frames)
    env = dict(os.environ, H264ENC_QP="30", H264ENC_KEYINT="25")
    a, b = tmp_path / "hip.264", tmp_path / "cpu.264"
    r = subprocess.run([HIPCLI, "-", str(w), str(h), str(a)], input=raw, capture_output=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    assert b"Frame     4:" in r.stdout
    r2 = subprocess.run([SCALAR, "-", str(w), str(h), str(b)], input=raw, capture_output=True, env=env, timeout=300)
    assert r2.returncode == 0
    assert a.read_bytes() == b.read_bytes()
    dec = _decode_all(a.read_bytes(), frames)
    assert dec.crop_size() == (w, h)
    # the header extensions through the same environment knobs on both command lines
    env = dict(env, H264ENC_AUD="1", H264ENC_FPS_NUM="25", H264ENC_REPEAT_HEADERS="1", H264ENC_KEYINT="2")
    assert subprocess.run([HIPCLI, "-", str(w), str(h), str(a)], input=raw, capture_output=True, env=env, timeout=300).returncode == 0
    assert subprocess.run([SCALAR, "-", str(w), str(h), str(b)], input=raw, capture_output=True, env=env, timeout=300).returncode == 0
    assert a.read_bytes() == b.read_bytes() and a.read_bytes()[:6] == bytes([0, 0, 0, 1, 9, 0x10])
    dec = _decode_all(a.read_bytes(), frames)
    assert dec.frame_rate() == (50, 2) and dec.aud_count() == frames


@pytest.mark.gpu
def test_hip_cli_usage_and_config_error(tmp_path):
    r = subprocess.run([HIPCLI, "-", "320"], capture_output=True)
    assert r.returncode == 255 and b"Usage:" in r.stdout
    r = subprocess.run([HIPCLI, "-", "321", "240", str(tmp_path / "o.264")], input=b"", capture_output=True)
    assert r.returncode != 0 and b"CEDAR_IOCTL_CONFIG failed" in r.stderr
    # malformed option strings are refused with their format, not half-applied
    for env in (dict(H264ENC_MAD="1,2,3:4"), dict(H264ENC_ROI="1,1,2,2,-4;3,3,x"), dict(H264ENC_ROI="0,0,1,1,1;" * 5)):
        r = subprocess.run([HIPCLI, "-", "64", "48", str(tmp_path / "o.264")], input=b"", capture_output=True, env=dict(os.environ, **env))
        assert r.returncode != 0 and (b"H264ENC_MAD wants" in r.stderr or b"H264ENC_ROI wants" in r.stderr), env


@pytest.mark.gpu
def test_hip_cli_skips_exactly_the_failed_frame(tmp_path):
    """h264enc.c:189-192: a failed ENCODE is reported, that frame is not written, the loop goes on and
    the driver's GOP counter and reference frames are as before it (cedar.c:1182-1201). The pipelined
    pump does the same with frames in flight: the file is the oracle's encode of the input WITHOUT the
    failed frame, the process ends with 0 like the reference. More frames than the session's ring holds,
    the failure past the first ring-full: the pump keeps the ring full, so the frame depth - 1 behind the
    failed one has reused the device slot of the frame before it."""
    w, h, frames, bad = 640, 368, 44, 21
    raw = _raw("B", w, h, frames)
    out = tmp_path / "out.264"
    env = dict(os.environ, H264ENC_QP="26", H264ENC_KEYINT="5", CEDAR_FAULT_INJECT="1", CEDAR_FAULT_INJECT_AT=str(bad + 1))
    r = subprocess.run([HIPCLI, "-", str(w), str(h), str(out)], input=raw, capture_output=True, env=env, timeout=200)
    assert r.returncode == 0
    assert b"CEDAR_IOCTL_ENCODE failed" in r.stderr and r.stderr.count(b"failed") == 1
    enc = ol.OracleEncoder(ol.default_config(w, h, qp=26, keyint=5))
    want = b"".join(enc.encode(*ol.gen_frame("B", w, h, f)) for f in range(frames) if f != bad)
    assert out.read_bytes() == want
    _decode_all(want, frames - 1)


@pytest.mark.gpu
def test_hip_cli_gop_parallel_subpel_and_first_frame(tmp_path):
    """H264ENC_GOP_PARALLEL (G closed GOPs side by side, flushed at EOF), H264ENC_SUBPEL and
    H264ENC_FIRST_FRAME (a GOP shard) through the command line: the bytes of the plain run / the oracle."""
    w, h, frames, K = 320, 240, 23, 4
    raw = _raw("B", w, h, frames)
    enc = ol.OracleEncoder(ol.default_config(w, h, qp=28, keyint=K, subpel=2))
    aus = [enc.encode(*ol.gen_frame("B", w, h, f)) for f in range(frames)]
    out = tmp_path / "gop.264"
    env = dict(os.environ, H264ENC_QP="28", H264ENC_KEYINT=str(K), H264ENC_SUBPEL="2", H264ENC_GOP_PARALLEL="3")
    r = subprocess.run([HIPCLI, "-", str(w), str(h), str(out)], input=raw + b"\1" * 999, capture_output=True, env=env, timeout=200)
    assert r.returncode == 0, r.stderr[-500:]
    assert out.read_bytes() == b"".join(aus)
    # the shard that starts at GOP 2: frames 8.. of the same input, no parameter sets (they go with frame 0)
    out2 = tmp_path / "shard.264"
    env2 = dict(os.environ, H264ENC_QP="28", H264ENC_KEYINT=str(K), H264ENC_SUBPEL="2", H264ENC_FIRST_FRAME=str(2 * K))
    fsz = w * h * 3 // 2
    r = subprocess.run([HIPCLI, "-", str(w), str(h), str(out2)], input=raw[2 * K * fsz:4 * K * fsz], capture_output=True, env=env2, timeout=200)
    assert r.returncode == 0, r.stderr[-500:]
    assert out2.read_bytes() == b"".join(aus[2 * K:4 * K])
    # rate control with GOP lanes (one QP per group of lanes): both command lines take the same two variables
    frames3, K3 = 110, 4
    raw3 = _raw("C", 192, 112, frames3)
    env3 = dict(os.environ, H264ENC_QP="30", H264ENC_KEYINT=str(K3), H264ENC_BITRATE="1480000", H264ENC_GOP_PARALLEL="3")
    a, b = tmp_path / "rc_hip.264", tmp_path / "rc_cpu.264"
    assert subprocess.run([HIPCLI, "-", "192", "112", str(a)], input=raw3, capture_output=True, env=env3, timeout=200).returncode == 0
    assert subprocess.run([SCALAR, "-", "192", "112", str(b)], input=raw3, capture_output=True, env=env3, timeout=200).returncode == 0
    assert a.read_bytes() == b.read_bytes()
    fixed = tmp_path / "fixed.264"
    assert subprocess.run([SCALAR, "-", "192", "112", str(fixed)], input=raw3, capture_output=True, env=dict(env3, H264ENC_BITRATE="0"), timeout=200).returncode == 0
    assert fixed.read_bytes() != b.read_bytes()   # (the law did act: 110 frames are 9 groups of 12, it sets in at the sixth)


@pytest.mark.gpu
def test_hip_cli_intra_refresh_and_regions_of_interest(tmp_path):
    """H264ENC_INTRA_REFRESH, H264ENC_ROI ("x,y,w,h,dqp;..." in macroblocks) and H264ENC_MAD ("t0,t1,t2,t3:d0,..,d4")
    through both command lines:
    the HIP one writes the file the scalar one writes."""
    w, h, frames = 320, 240, 12
    raw = _raw("B", w, h, frames)
    src = tmp_path / "in.nv12"
    src.write_bytes(raw)
    env = dict(os.environ, H264ENC_QP="29", H264ENC_KEYINT="25", H264ENC_INTRA_REFRESH="5",
               H264ENC_ROI="2,1,6,4,-8;0,0,20,15,6", H264ENC_SUBPEL="1", H264ENC_MAD="1,3,6,12:-3,-1,0,2,4")
    out_c, out_g = tmp_path / "cpu.264", tmp_path / "gpu.264"
    r = subprocess.run([SCALAR, str(src), str(w), str(h), str(out_c)], capture_output=True, env=env, timeout=200)
    assert r.returncode == 0, r.stderr[-500:]
    r = subprocess.run([HIPCLI, str(src), str(w), str(h), str(out_g)], capture_output=True, env=env, timeout=200)
    assert r.returncode == 0, r.stderr[-500:]
    assert out_g.read_bytes() == out_c.read_bytes() and len(out_c.read_bytes()) > 1000
    enc = ol.OracleEncoder(ol.default_config(w, h, qp=29, keyint=25, intra_refresh=5, subpel=1,
                                             roi=[(2, 1, 6, 4, -8), (0, 0, 20, 15, 6)], mad_th=[1, 3, 6, 12], mad_dqp=[-3, -1, 0, 2, 4]))
    assert out_c.read_bytes() == b"".join(enc.encode(*ol.gen_frame("B", w, h, f)) for f in range(frames))
