"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and
exports every symbol include/cedar_h264enc.h declares; without a GPU it refuses to
open (no CPU fallback); its VLC tables equal the oracle's. No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import cedarx_h264_encoder_amd as m
    from cedarx_h264_encoder_amd import build
    build.build()
    return m.load_library()


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "cedar_h264enc.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cedar_enc_\w+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), n


def test_reference_verbs_present():
    names = _declared_functions()
    for verb in ("cedar_enc_open", "cedar_enc_config", "cedar_enc_encode", "cedar_enc_close"):
        assert verb in names


def test_config_struct_mirrors_reference_field_order():
    """struct cedar_enc_config keeps the field order of struct cedar_ioctl_config
    (kernel/cedar_ioctl.h:12-46) with *_dma_addr turned into pointers."""
    src = open(os.path.join(ROOT, "include", "cedar_h264enc.h")).read()
    body = re.search(r"struct cedar_enc_config \{(.*?)\n\};", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"(\w+);", body)
    assert fields == ["src_width", "src_height", "src_format", "dst_width", "dst_height", "profile", "level",
                      "qp", "keyframe_interval", "thumbnail", "thumbnail_downscale", "entropy_coding_mode",
                      "input_luma", "input_luma_size", "input_chroma", "input_chroma_size",
                      "bytestream", "bytestream_size", "thumb_luma", "thumb_luma_size",
                      "thumb_chroma", "thumb_chroma_size"]


def test_open_fails_loudly_without_gpu(lib):
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    import cedarx_h264_encoder_amd as m
    with pytest.raises(m.EncoderError) as e:
        m.CedarEncoder(width=64, height=48)
    assert e.value.errno == 19  # ENODEV


def test_product_never_references_the_oracle():
    """The oracle is test infrastructure: nothing under the package may mention it."""
    pkg = os.path.join(ROOT, "cedarx_h264_encoder_amd")
    for d, _, files in os.walk(pkg):
        if os.sep + "lib" in d:
            continue
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                assert "h264o" not in txt and "oracle/" not in txt and "oracle_lib" not in txt, f


def test_product_vlc_tables_equal_oracle_tables():
    from tests.test_tables import T
    src = open(os.path.join(ROOT, "cedarx_h264_encoder_amd", "csrc", "enc_tables.h")).read()

    def arr(name):
        m = re.search(name + r"\[\d+\] = \{(.*?)\};", src, re.S)
        return [int(x, 0) for x in re.findall(r"0x[0-9a-fA-F]+|\d+", m.group(1))]

    def pack(l, b):
        return [(int(x) << 8) | int(y) for x, y in zip(l, b)]

    ct = arr("kCoeffToken")
    for t in range(4):
        assert ct[68 * t:68 * t + 68] == pack(T["h264o_coeff_token_len"][t], T["h264o_coeff_token_bits"][t])
    assert arr("kCoeffTokenChromaDC") == pack(T["h264o_chroma_dc_coeff_token_len"], T["h264o_chroma_dc_coeff_token_bits"])
    tz = arr("kTotalZeros")
    for t in range(15):
        assert tz[16 * t:16 * t + 16] == pack(T["h264o_total_zeros_len"][t], T["h264o_total_zeros_bits"][t])
    rb = arr("kRunBefore")
    for t in range(7):
        assert rb[16 * t:16 * t + 15] == pack(T["h264o_run_before_len"][t], T["h264o_run_before_bits"][t])
    ci, cp = arr("kCbpCodeIntra"), arr("kCbpCodeInter")
    for code in range(48):
        assert ci[int(T["h264o_cbp_intra_from_code"][code])] == code
        assert cp[int(T["h264o_cbp_inter_from_code"][code])] == code


def test_pattern_generators_agree():
    """bench/CLI generator (product, host C++) vs the oracle's generator."""
    import cedarx_h264_encoder_amd as m
    from tests import oracle_lib as ol
    for p in "ABC":
        for f in (0, 3):
            a = m.gen_pattern(p, 70, 50, f, 5)
            b = ol.gen_frame(p, 70, 50, f, 5)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_ctypes_mirrors_have_the_layout_of_the_c_header(tmp_path):
    """encoder.py's _ConfigExt / _Config / _Stats against include/cedar_h264enc.h as gcc lays it out: size and the
    offset of every field (the library refuses a config_ext of another size, but only on a GPU box)."""
    import ctypes
    import subprocess
    from cedarx_h264_encoder_amd import encoder as enc
    src = tmp_path / "layout.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "cedar_h264enc.h"', 'int main(void) {']
    structs = (("cedar_enc_config_ext", enc._ConfigExt), ("cedar_enc_config", enc._Config), ("cedar_enc_stats", enc._Stats))
    for cname, ct in structs:
        lines.append('printf("%s %%zu\\n", sizeof(struct %s));' % (cname, cname))
        for f in ct._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(struct %s, %s));' % (cname, f[0], cname, f[0]))
    lines += ['return 0; }']
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    got = dict(ln.split() for ln in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for cname, ct in structs:
        assert int(got[cname]) == ctypes.sizeof(ct), cname
        for f in ct._fields_:
            assert int(got["%s.%s" % (cname, f[0])]) == getattr(ct, f[0]).offset, (cname, f[0])


def test_every_environment_switch_the_library_reads_is_documented():
    """INTEGRATION.md lists the CEDAR_* switches; the library must not read one it does not name, nor the document name one
    that nothing reads."""
    import glob
    doc = set(re.findall(r"CEDAR_[A-Z0-9_]+", open(os.path.join(ROOT, "INTEGRATION.md")).read()))
    src = ""
    for f in glob.glob(os.path.join(ROOT, "cedarx_h264_encoder_amd", "csrc", "*")):
        if f.endswith((".cpp", ".hip", ".h", ".c")):
            src += open(f).read()
    read = set(re.findall(r'getenv\("(CEDAR_[A-Z0-9_]+)"\)', src))
    assert not sorted(read - doc), "read but not documented"
    everywhere = set(re.findall(r"CEDAR_[A-Z0-9_]+", src + open(os.path.join(ROOT, "include", "cedar_h264enc.h")).read() +
                                open(os.path.join(ROOT, "bench.py")).read() + open(os.path.join(ROOT, "cedarx_h264_encoder_amd", "encoder.py")).read()))
    assert not sorted(doc - everywhere), "documented but unknown to the sources"
