"""The scalar encoder and the test decoder under AddressSanitizer + UBSan (CPU build only:
the GPU pool runs no sanitizers). A clip with ragged geometry, several slices and both
picture types goes through both tools; any report makes the tool exit non-zero."""
import os
import subprocess

import numpy as np
import pytest

from tests import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle")


@pytest.fixture(scope="module")
def san_tools():
    r = subprocess.run(["make", "-C", ORACLE, "san"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build unavailable: " + r.stderr[-300:])
    return os.path.join(ORACLE, "build", "h264enc_scalar_san"), os.path.join(ORACLE, "build", "h264dec_san")


@pytest.mark.parametrize("w,h,slices,qp", [(70, 50, 1, 24), (176, 144, 3, 30), (64, 48, 2, 47)])
def test_scalar_encoder_and_decoder_are_sanitizer_clean(san_tools, tmp_path, w, h, slices, qp):
    enc, dec = san_tools
    clip = b"".join(np.concatenate(ol.gen_frame("C", w, h, f)).tobytes() for f in range(5))
    out = tmp_path / "o.264"
    env = dict(os.environ, H264ENC_QP=str(qp), H264ENC_KEYINT="3", H264ENC_SLICES=str(slices),
               ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([enc, "-", str(w), str(h), str(out)], input=clip, capture_output=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert b"runtime error" not in r.stderr and b"AddressSanitizer" not in r.stderr, r.stderr.decode()[-2000:]
    # same bytes as the regular build of the library
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=3, slice_count=slices))
    want = b"".join(oe.encode(*ol.gen_frame("C", w, h, f)) for f in range(5))
    assert out.read_bytes() == want
    r = subprocess.run([dec, str(out), str(tmp_path / "o.nv12")], capture_output=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert b"runtime error" not in r.stderr and b"AddressSanitizer" not in r.stderr, r.stderr.decode()[-2000:]
    assert os.path.getsize(tmp_path / "o.nv12") >= 5 * w * h * 3 // 2


def test_sub_sample_rate_control_and_bit_cap_paths_are_sanitizer_clean(san_tools, tmp_path):
    """round 2's options through the sanitizer build: quarter-sample refinement and interpolation (encoder and
    decoder), the rate control law, and the requantisation of pictures with oversize macroblocks."""
    enc, dec = san_tools
    w, h, n = 96, 80, 22
    clip = b"".join(np.concatenate(ol.gen_frame("CB"[f % 3 > 0], w, h, f)).tobytes() for f in range(n))
    out = tmp_path / "o.264"
    env = dict(os.environ, H264ENC_QP="6", H264ENC_KEYINT="7", H264ENC_SUBPEL="2", H264ENC_BITRATE="900000",
               H264ENC_STRICT_MB_BITS="2", H264ENC_PCM="-1", ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([enc, "-", str(w), str(h), str(out)], input=clip, capture_output=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert b"runtime error" not in r.stderr and b"AddressSanitizer" not in r.stderr, r.stderr.decode()[-2000:]
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=6, keyint=7, subpel=2, bitrate=900000, strict_mb_bits=2, pcm=-1))
    want = b"".join(oe.encode(*ol.gen_frame("CB"[f % 3 > 0], w, h, f)) for f in range(n))
    assert out.read_bytes() == want
    r = subprocess.run([dec, str(out), str(tmp_path / "o.nv12")], capture_output=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert b"runtime error" not in r.stderr and b"AddressSanitizer" not in r.stderr, r.stderr.decode()[-2000:]


def test_group_rate_control_path_is_sanitizer_clean(san_tools, tmp_path):
    """round 3: the rate control law of GOP lanes (one QP per group, h264o_config.rc_lanes; H264ENC_GOP_PARALLEL on the
    scalar command line) through the sanitizer build, long enough for the law to act and for its ring of groups to be used."""
    enc, dec = san_tools
    w, h, n, K, G = 64, 48, 150, 2, 2
    clip = b"".join(np.concatenate(ol.gen_frame("CB"[(f // 5) % 2], w, h, f)).tobytes() for f in range(n))
    out = tmp_path / "o.264"
    env = dict(os.environ, H264ENC_QP="28", H264ENC_KEYINT=str(K), H264ENC_BITRATE="400000", H264ENC_GOP_PARALLEL=str(G),
               ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([enc, "-", str(w), str(h), str(out)], input=clip, capture_output=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert b"runtime error" not in r.stderr and b"AddressSanitizer" not in r.stderr, r.stderr.decode()[-2000:]
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=28, keyint=K, bitrate=400000, rc_lanes=G))
    want, qps = b"", set()
    for f in range(n):
        want += oe.encode(*ol.gen_frame("CB"[(f // 5) % 2], w, h, f))
        qps.add(oe.frame_qp())
    assert out.read_bytes() == want and len(qps) > 1     # 37 groups of 4 frames: the ring of 32 wraps, the law sets in at group 9
    r = subprocess.run([dec, str(out), str(tmp_path / "o.nv12")], capture_output=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert b"runtime error" not in r.stderr and b"AddressSanitizer" not in r.stderr, r.stderr.decode()[-2000:]
