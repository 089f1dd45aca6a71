"""ctypes binding of the CPU oracle (oracle/build/libh264o.so) for the tests.

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this module. The product package never does.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB_PATH = os.path.join(ROOT, "oracle", "build", "libh264o.so")

MB_I4x4, MB_I16x16, MB_P16x16, MB_PSKIP, MB_IPCM = 0, 1, 2, 3, -1


class Config(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "width", "height", "qp", "keyint", "profile_idc", "level_idc",
        "search_range", "slice_count", "chroma_qp_offset", "repeat_headers", "aud", "fps_num", "fps_den",
        "subpel", "bitrate", "strict_mb_bits", "intra_refresh", "roi_count")] + [
        ("roi", (ctypes.c_int * 5) * 4), ("mad_th", ctypes.c_int * 4), ("mad_dqp", ctypes.c_int * 5), ("scene_cut", ctypes.c_int), ("pcm", ctypes.c_int), ("rc_lanes", ctypes.c_int)]


class MB(ctypes.Structure):
    _fields_ = [
        ("type", ctypes.c_int8), ("i16_mode", ctypes.c_int8),
        ("chroma_mode", ctypes.c_int8), ("cbp", ctypes.c_int8),
        ("i4_modes", ctypes.c_int8 * 16),
        ("mvx", ctypes.c_int16), ("mvy", ctypes.c_int16),
        ("nnz", ctypes.c_uint8 * 24),
        ("luma", (ctypes.c_int16 * 16) * 16),
        ("dc16", ctypes.c_int16 * 16),
        ("cdc", (ctypes.c_int16 * 4) * 2),
        ("cac", ((ctypes.c_int16 * 16) * 4) * 2),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
        L = ctypes.CDLL(_LIB_PATH)
        vp, ci = ctypes.c_void_p, ctypes.c_int
        L.h264o_default_config.argtypes = [ctypes.POINTER(Config), ci, ci]
        L.h264o_open.restype = vp
        L.h264o_open.argtypes = [ctypes.POINTER(Config), ctypes.POINTER(ci)]
        L.h264o_close.argtypes = [vp]
        L.h264o_encode.argtypes = [vp, vp, vp, vp, ci]
        L.h264o_max_frame_bytes.argtypes = [vp]
        L.h264o_mb_width.argtypes = [vp]
        L.h264o_mb_height.argtypes = [vp]
        L.h264o_mbs.restype = ctypes.POINTER(MB)
        L.h264o_mbs.argtypes = [vp]
        L.h264o_recon_luma.restype = vp
        L.h264o_recon_luma.argtypes = [vp]
        L.h264o_recon_chroma.restype = vp
        L.h264o_recon_chroma.argtypes = [vp]
        L.h264o_last_frame_type.argtypes = [vp]
        L.h264o_max_mb_bits.argtypes = [vp]
        L.h264o_frame_qp.argtypes = [vp]
        L.h264o_write_sps.argtypes = [ctypes.POINTER(Config), vp, ci]
        L.h264o_write_pps.argtypes = [ctypes.POINTER(Config), vp, ci]
        L.h264o_write_aud.argtypes = [ci, vp, ci]
        L.h264o_write_slice_header.argtypes = [ctypes.POINTER(Config), ci, ci, ci, ci, vp, ci,
                                               ctypes.POINTER(ci)]
        L.h264o_cavlc_block.argtypes = [vp, ci, ci, vp, ci]
        L.h264o_fdct4x4.argtypes = [vp, vp]
        L.h264o_idct4x4_add.argtypes = [vp, vp, ci]
        L.h264o_ue_len.argtypes = [ctypes.c_uint]
        L.h264o_gen_frame.argtypes = [ci, ci, ci, ci, ctypes.c_uint, vp, vp]
        L.h264o_dec_open.restype = vp
        L.h264o_dec_close.argtypes = [vp]
        L.h264o_dec_decode.argtypes = [vp, vp, ci]
        for n in ("width", "height", "crop_width", "crop_height"):
            getattr(L, "h264o_dec_" + n).argtypes = [vp]
        L.h264o_dec_frame_rate.argtypes = [vp, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint)]
        L.h264o_dec_frame_rate.restype = None
        L.h264o_dec_reorder_frames.argtypes = [vp]
        L.h264o_dec_aud_count.argtypes = [vp]
        L.h264o_dec_luma.restype = vp
        L.h264o_dec_luma.argtypes = [vp]
        L.h264o_dec_chroma.restype = vp
        L.h264o_dec_chroma.argtypes = [vp]
        L.h264o_dec_error.restype = ctypes.c_char_p
        L.h264o_dec_error.argtypes = [vp]
        _lib = L
    return _lib


# I_PCM in the twin's sessions unless a test says otherwise (h264o_config.pcm: 0 on, -1 off)
PCM_DEFAULT = 0


def default_config(width, height, **kw):
    c = Config()
    lib().h264o_default_config(ctypes.byref(c), width, height)
    c.pcm = PCM_DEFAULT
    for k, v in kw.items():
        if k == "roi":   # [(x, y, w, h, qp_offset), ...] in macroblocks
            c.roi_count = len(v)
            for i, r in enumerate(v):
                for j in range(5):
                    c.roi[i][j] = int(r[j])
        elif k in ("mad_th", "mad_dqp"):
            for i, x in enumerate(v):
                getattr(c, k)[i] = int(x)
        else:
            setattr(c, k, v)
    return c


def gen_frame(pattern, width, height, frame, seed_offset=0):
    """Synthetic nv12 frame of SURVEY.md 8(d): returns (luma[h*w], chroma[h*w/2])."""
    y = np.zeros(width * height, np.uint8)
    uv = np.zeros(width * height // 2, np.uint8)
    lib().h264o_gen_frame(ord(pattern), width, height, frame, seed_offset,
                          y.ctypes.data, uv.ctypes.data)
    return y, uv


def _view(ptr, shape):
    return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape)


class OracleEncoder:
    def __init__(self, cfg):
        err = ctypes.c_int()
        self.cfg = cfg
        self.h = lib().h264o_open(ctypes.byref(cfg), ctypes.byref(err))
        if not self.h:
            raise OSError(-err.value, os.strerror(-err.value))
        self.cap = lib().h264o_max_frame_bytes(self.h)
        self.out = np.zeros(self.cap, np.uint8)
        self.cw = (cfg.width + 15) & ~15
        self.ch = (cfg.height + 15) & ~15

    def encode(self, luma, chroma):
        luma = np.ascontiguousarray(luma, np.uint8)
        chroma = np.ascontiguousarray(chroma, np.uint8)
        assert luma.size == self.cfg.width * self.cfg.height
        assert chroma.size == self.cfg.width * self.cfg.height // 2
        n = lib().h264o_encode(self.h, luma.ctypes.data, chroma.ctypes.data,
                               self.out.ctypes.data, self.cap)
        if n < 0:
            raise OSError(-n, os.strerror(-n))
        return self.out[:n].tobytes()

    def frame_qp(self):
        """slice QP of the last frame (cfg.qp unless rate control is on)"""
        return lib().h264o_frame_qp(self.h)

    def mb_qp(self):
        """QP_Y the decoder holds for every macroblock of the last frame (regions of interest)"""
        L = lib()
        L.h264o_mb_qp.restype = ctypes.POINTER(ctypes.c_uint8)
        L.h264o_mb_qp.argtypes = [ctypes.c_void_p]
        n = L.h264o_mb_width(self.h) * L.h264o_mb_height(self.h)
        return np.ctypeslib.as_array(L.h264o_mb_qp(self.h), shape=(n,)).copy()

    def max_mb_bits(self):
        """largest macroblock_layer() of the last frame, in bits (H.264 A.3.1 allows 3200)"""
        return lib().h264o_max_mb_bits(self.h)

    def recon(self):
        y = _view(lib().h264o_recon_luma(self.h), (self.ch, self.cw)).copy()
        uv = _view(lib().h264o_recon_chroma(self.h), (self.ch // 2, self.cw)).copy()
        return y, uv

    def mbs(self):
        n = lib().h264o_mb_width(self.h) * lib().h264o_mb_height(self.h)
        p = lib().h264o_mbs(self.h)
        return [p[i] for i in range(n)]

    def mb_array(self):
        """All MB records as one structured numpy array (copy)."""
        n = lib().h264o_mb_width(self.h) * lib().h264o_mb_height(self.h)
        p = lib().h264o_mbs(self.h)
        buf = ctypes.string_at(p, n * ctypes.sizeof(MB))
        dt = np.dtype([("type", "i1"), ("i16_mode", "i1"), ("chroma_mode", "i1"), ("cbp", "i1"),
                       ("i4_modes", "i1", 16), ("mvx", "<i2"), ("mvy", "<i2"), ("nnz", "u1", 24),
                       ("luma", "<i2", (16, 16)), ("dc16", "<i2", 16), ("cdc", "<i2", (2, 4)),
                       ("cac", "<i2", (2, 4, 16))])
        assert dt.itemsize == ctypes.sizeof(MB)
        return np.frombuffer(buf, dt).copy()

    def close(self):
        if self.h:
            lib().h264o_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OracleDecoder:
    def __init__(self):
        self.h = lib().h264o_dec_open()

    def decode(self, data):
        buf = np.frombuffer(data, np.uint8)
        r = lib().h264o_dec_decode(self.h, buf.ctypes.data, buf.size)
        if r < 0:
            raise ValueError("decode error: " + lib().h264o_dec_error(self.h).decode())
        return r

    def picture(self):
        cw, ch = lib().h264o_dec_width(self.h), lib().h264o_dec_height(self.h)
        y = _view(lib().h264o_dec_luma(self.h), (ch, cw)).copy()
        uv = _view(lib().h264o_dec_chroma(self.h), (ch // 2, cw)).copy()
        return y, uv

    def crop_size(self):
        return lib().h264o_dec_crop_width(self.h), lib().h264o_dec_crop_height(self.h)

    def frame_rate(self):
        """(num, den) from the SPS VUI; (0, 0) without timing."""
        n, d = ctypes.c_uint(), ctypes.c_uint()
        lib().h264o_dec_frame_rate(self.h, ctypes.byref(n), ctypes.byref(d))
        return n.value, d.value

    def reorder_frames(self):
        return lib().h264o_dec_reorder_frames(self.h)

    def aud_count(self):
        return lib().h264o_dec_aud_count(self.h)

    def close(self):
        if self.h:
            lib().h264o_dec_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
