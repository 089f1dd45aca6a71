"""ctypes binding of libcedar_h264enc.so (include/cedar_h264enc.h).

Names follow the reference's interface: the config fields are those of
struct cedar_ioctl_config (kernel/cedar_ioctl.h:12-46), `config()` is
CEDAR_IOCTL_CONFIG (cedar.c:732-866) and `encode()` is CEDAR_IOCTL_ENCODE
(cedar.c:1032-1209) with the same return convention: bytes produced, or an
OSError carrying the errno the ioctl would have set.
"""
import ctypes
import os
from dataclasses import dataclass

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_PKG, "lib", "libcedar_h264enc.so")

FORMAT_NV12 = 0
ENTROPY_CAVLC = 0
ENTROPY_CABAC = 1
STAGES = ("input", "intra_decide", "intra_recon", "inter", "deblock", "entropy")
DBG = {"mbinfo": 0, "i4modes": 1, "nnz": 2, "coef": 3, "recon_y": 4, "recon_uv": 5, "src_y": 6,
       "src_uv": 7, "mb_bits": 8, "mb_qp": 9}


class EncoderError(OSError):
    pass


class _Config(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "src_width", "src_height", "src_format", "dst_width", "dst_height", "profile", "level",
        "qp", "keyframe_interval", "thumbnail", "thumbnail_downscale", "entropy_coding_mode")] + [
        ("input_luma", ctypes.c_void_p), ("input_luma_size", ctypes.c_int),
        ("input_chroma", ctypes.c_void_p), ("input_chroma_size", ctypes.c_int),
        ("bytestream", ctypes.c_void_p), ("bytestream_size", ctypes.c_int),
        ("thumb_luma", ctypes.c_void_p), ("thumb_luma_size", ctypes.c_int),
        ("thumb_chroma", ctypes.c_void_p), ("thumb_chroma_size", ctypes.c_int)]


class _ConfigExt(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "size", "device", "n_streams", "slice_count", "slice_first", "slice_num", "search_range",
        "chroma_qp_offset_minus4", "repeat_headers", "aud", "fps_num", "fps_den", "first_frame_index",
        "gop_parallel", "strict_mb_bits", "subpel", "gop_low_latency", "bitrate", "intra_refresh", "roi_count")] + [
        ("roi", (ctypes.c_int * 5) * 4), ("mad_th", ctypes.c_int * 4), ("mad_dqp", ctypes.c_int * 5), ("pcm", ctypes.c_int), ("scene_cut", ctypes.c_int)]


class _Stats(ctypes.Structure):
    _fields_ = [("frames", ctypes.c_uint64), ("launches", ctypes.c_uint64 * 6),
                ("ms", ctypes.c_double * 6), ("encode_wall_ms", ctypes.c_double),
                ("wf_split", ctypes.c_int), ("deblock_planes_split", ctypes.c_int),
                ("frames_pipelined", ctypes.c_int), ("idr_lookahead", ctypes.c_int), ("depth", ctypes.c_int),
                ("deblock_streams", ctypes.c_int), ("oversize_frames", ctypes.c_uint64), ("frame_qp", ctypes.c_int),
                ("requantised_frames", ctypes.c_uint64), ("scene_cuts", ctypes.c_uint64), ("deblock_rows", ctypes.c_int), ("intra_rows", ctypes.c_int), ("src_direct", ctypes.c_int),
                ("entropy_fused", ctypes.c_int)]


_lib = None


def library_path():
    return _LIB


def _share_hip_runtime_with_torch():
    """One process must hold ONE HIP runtime. PyTorch-ROCm ships its own
    libamdhip64.so.7 (same soname as /opt/rocm's); if this library pulled in the
    system copy first, a later `import torch` would load a second runtime and
    see no GPU. So when torch is installed (bench.py and the tests use it for
    torch.distributed / device buffers) its copy is loaded first and the dynamic
    linker binds ours to it by soname. The C CLI (no torch) uses /opt/rocm."""
    import sys
    if "torch" in sys.modules or os.environ.get("CEDAR_H264ENC_SYSTEM_HIP"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        hip = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(hip):
            ctypes.CDLL(hip, mode=ctypes.RTLD_GLOBAL)
    except OSError:
        pass


# Frame pipelining wants the deblock streams on hardware queues of their own; the HIP runtime
# reads this when it starts (first HIP call of the process), the session probes the result.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def load_library():
    """Load the native library; build it first if it is missing (hipcc is
    present on the GPU boxes). Fails loudly: there is no other implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        from . import build as _build
        _build.build()
    _share_hip_runtime_with_torch()
    L = ctypes.CDLL(_LIB)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.cedar_enc_open.restype = vp
    L.cedar_enc_config.argtypes = [vp, ctypes.POINTER(_Config)]
    L.cedar_enc_config_ext.argtypes = [vp, ctypes.POINTER(_Config), ctypes.POINTER(_ConfigExt)]
    L.cedar_enc_encode.argtypes = [vp]
    L.cedar_enc_pipeline_depth.argtypes = []
    L.cedar_enc_pipeline_depth.restype = ctypes.c_int
    L.cedar_enc_session_depth.argtypes = [vp]
    L.cedar_enc_submit.argtypes = [vp]
    L.cedar_enc_collect.argtypes = [vp]
    L.cedar_enc_flush.argtypes = [vp]
    L.cedar_enc_set_frame_index.argtypes = [vp, ci]
    L.cedar_enc_close.argtypes = [vp]
    L.cedar_enc_close.restype = None
    L.cedar_enc_stream_output.argtypes = [vp, ci, ctypes.POINTER(vp), ctypes.POINTER(ci)]
    L.cedar_enc_set_device_input.argtypes = [vp, ci, vp, vp]
    L.cedar_enc_set_profiling.argtypes = [vp, ci]
    L.cedar_enc_get_stats.argtypes = [vp, ctypes.POINTER(_Stats)]
    L.cedar_enc_reset_stats.argtypes = [vp]
    L.cedar_enc_debug_read.argtypes = [vp, ci, ci, vp, ci]
    L.cedar_enc_gen_pattern.argtypes = [ci, ci, ci, ci, ctypes.c_uint, vp, vp]
    L.cedar_enc_gen_pattern.restype = None
    L.cedar_enc_last_error.argtypes = [vp]
    L.cedar_enc_last_error.restype = ctypes.c_char_p
    L.cedar_enc_version.restype = ctypes.c_char_p
    _lib = L
    return L


def gen_pattern(pattern, width, height, frame, seed_offset=0):
    """Synthetic nv12 frame (SURVEY.md 8(d) patterns 'A', 'B', 'C')."""
    y = np.empty(width * height, np.uint8)
    uv = np.empty(width * height // 2, np.uint8)
    load_library().cedar_enc_gen_pattern(ord(pattern), width, height, frame, seed_offset,
                                         y.ctypes.data, uv.ctypes.data)
    return y, uv


@dataclass
class EncoderConfig:
    """Fields of struct cedar_ioctl_config (kernel/cedar_ioctl.h:12-46) plus the
    extension block; defaults are the reference CLI's (userspace/h264enc.c:53-66)
    except profile 66 / CAVLC, which BASELINE.json's north_star fixes."""
    width: int
    height: int
    qp: int = 24
    keyframe_interval: int = 25
    profile: int = 66
    level: int = 0            # 0 = pick 41, or 51 above 8192 macroblocks
    entropy_coding_mode: int = ENTROPY_CAVLC
    src_format: int = FORMAT_NV12
    dst_width: int = 0        # 0 = ALIGN(width, 16) as the reference CLI does
    dst_height: int = 0
    # extensions
    device: int = 0
    n_streams: int = 1
    slice_count: int = 1
    slice_first: int = 0
    slice_num: int = 0
    search_range: int = 0
    chroma_qp_offset_minus4: int = 0
    repeat_headers: int = 0
    aud: int = 0              # 1: access unit delimiter in front of every access unit
    fps_num: int = 0          # both > 0: VUI timing + bitstream restriction in the SPS
    fps_den: int = 0
    first_frame_index: int = 0  # global index of this session's first frame (a multiple of keyframe_interval)
    gop_parallel: int = 0       # G > 1: G consecutive closed GOPs of every stream are encoded side by side
    strict_mb_bits: int = 0     # a picture with a macroblock above 3200 bits: 0 counted, 1 EOVERFLOW, 2 coded again at QP + 6
    subpel: int = 0             # motion vectors: 0 whole samples, 1 half, 2 quarter
    gop_low_latency: int = 0    # gop_parallel: 1 = half the latency, every group's IDR step exposed
    bitrate: int = 0            # > 0: frame-level rate control towards this many bits per second
    intra_refresh: int = 0      # N >= 3: macroblocks with (x + 2*y + t) % N == 0 of the t-th P picture are coded intra
    roi: tuple = ()             # up to 4 regions of interest: (x, y, width, height in macroblocks, QP offset)
    mad_th: tuple = (0, 0, 0, 0)       # MAD classes: ascending thresholds on luma SAD / 256 against the chosen prediction
    mad_dqp: tuple = (0, 0, 0, 0, 0)   # ... and the QP offset of each of the five classes (all zero: off)
    pcm: int = 0                # 0: macroblocks that could pass 3200 bits are coded I_PCM (where the session takes it); -1: never
    scene_cut: int = 0          # 1..100: a P picture whose macroblocks vote for it by this percentage becomes an IDR picture


class CedarEncoder:
    """One encode session == one open of /dev/cedar_dev in the reference."""

    def __init__(self, cfg=None, **kw):
        self._L = load_library()
        self._h = self._L.cedar_enc_open()
        if not self._h:
            raise EncoderError(19, "cedar_enc_open: no usable HIP device (no CPU fallback exists)")
        self.cfg = None
        if cfg is not None or kw:
            self.config(cfg if cfg is not None else EncoderConfig(**kw))

    # -- CEDAR_IOCTL_CONFIG --
    def config(self, cfg):
        c = _Config()
        c.src_width, c.src_height, c.src_format = cfg.width, cfg.height, cfg.src_format
        c.dst_width = cfg.dst_width or (cfg.width + 15) & ~15
        c.dst_height = cfg.dst_height or (cfg.height + 15) & ~15
        c.profile = cfg.profile
        c.level = cfg.level or (51 if (c.dst_width // 16) * (c.dst_height // 16) > 8192 else 41)
        c.qp, c.keyframe_interval = cfg.qp, cfg.keyframe_interval
        c.entropy_coding_mode = cfg.entropy_coding_mode
        x = _ConfigExt()
        x.size = ctypes.sizeof(_ConfigExt)
        for n in ("device", "n_streams", "slice_count", "slice_first", "slice_num", "search_range",
                  "chroma_qp_offset_minus4", "repeat_headers", "aud", "fps_num", "fps_den", "first_frame_index",
                  "gop_parallel", "strict_mb_bits", "subpel", "gop_low_latency", "bitrate", "intra_refresh", "pcm", "scene_cut"):
            setattr(x, n, getattr(cfg, n))
        if len(cfg.roi) > 4:
            raise EncoderError(22, "at most 4 regions of interest")
        x.roi_count = len(cfg.roi)
        for i, v in enumerate(cfg.mad_th):
            x.mad_th[i] = int(v)
        for i, v in enumerate(cfg.mad_dqp):
            x.mad_dqp[i] = int(v)
        for i, r in enumerate(cfg.roi):
            for j in range(5):
                x.roi[i][j] = int(r[j])
        r = self._L.cedar_enc_config_ext(self._h, ctypes.byref(c), ctypes.byref(x))
        self._check(r, "CEDAR_IOCTL_CONFIG")
        self.cfg = cfg
        self._c = c
        self.n_streams = max(1, cfg.n_streams)
        self.cw, self.ch = c.dst_width, c.dst_height
        self.mb_w, self.mb_h = self.cw // 16, self.ch // 16
        S = self.n_streams
        # the three "mmap"ed regions (h264enc.c:76-106), as numpy views of the pinned buffers
        self.input_luma = np.ctypeslib.as_array(
            ctypes.cast(c.input_luma, ctypes.POINTER(ctypes.c_uint8)), (S, c.input_luma_size))
        self.input_chroma = np.ctypeslib.as_array(
            ctypes.cast(c.input_chroma, ctypes.POINTER(ctypes.c_uint8)), (S, c.input_chroma_size))
        self.bytestream_size = c.bytestream_size
        return self

    def _check(self, r, what):
        if r < 0:
            msg = self._L.cedar_enc_last_error(self._h).decode()
            raise EncoderError(-r, "%s failed: %s (%s)" % (what, os.strerror(-r), msg))
        return r

    def set_input(self, luma, chroma, stream=0):
        """Write one frame into the input buffers (what read_frame() does, h264enc.c:182-187)."""
        w, h = self.cfg.width, self.cfg.height
        self.input_luma[stream, :w * h] = np.asarray(luma, np.uint8).reshape(-1)
        self.input_chroma[stream, :w * h // 2] = np.asarray(chroma, np.uint8).reshape(-1)

    def set_device_input(self, stream, d_luma_ptr, d_chroma_ptr):
        self._check(self._L.cedar_enc_set_device_input(self._h, stream, d_luma_ptr, d_chroma_ptr),
                    "set_device_input")

    # -- CEDAR_IOCTL_ENCODE --
    def encode(self):
        return self._check(self._L.cedar_enc_encode(self._h), "CEDAR_IOCTL_ENCODE")

    @property
    def pipeline_depth(self):
        """Frames this session allows in flight between submit() and collect()."""
        return self._check(self._L.cedar_enc_session_depth(self._h), "session_depth")

    def submit(self):
        return self._check(self._L.cedar_enc_submit(self._h), "submit")

    def collect(self):
        return self._check(self._L.cedar_enc_collect(self._h), "collect")

    def set_frame_index(self, index):
        """the next frame submitted is frame `index` of the stream and starts a GOP (GOP shards)"""
        return self._check(self._L.cedar_enc_set_frame_index(self._h, index), "set_frame_index")

    def flush(self):
        """gop_parallel sessions: no more frames will come, deliver what is buffered."""
        return self._check(self._L.cedar_enc_flush(self._h), "flush")

    def output(self, stream=0):
        p, n = ctypes.c_void_p(), ctypes.c_int()
        self._check(self._L.cedar_enc_stream_output(self._h, stream, ctypes.byref(p), ctypes.byref(n)),
                    "stream_output")
        return ctypes.string_at(p.value, n.value) if n.value else b""

    def encode_frame(self, luma, chroma, stream=0):
        """Convenience for one stream: set input, encode, return the access unit."""
        self.set_input(luma, chroma, stream)
        self.encode()
        return self.output(stream)

    # -- instrumentation --
    def set_profiling(self, on=True, stages=None, every=1):
        """HIP-event timing of the stages: all of them, or only the named ones (cheaper), on every
        frame or on every `every`-th one (cheaper still)."""
        sample = (min(max(int(every), 1), 255) << 8) if on else 0
        if on and stages:
            mask = 0
            for name in stages:
                mask |= 1 << (1 + STAGES.index(name))
            self._L.cedar_enc_set_profiling(self._h, mask | sample)
        else:
            self._L.cedar_enc_set_profiling(self._h, (1 if on else 0) | sample)

    def stats(self):
        s = _Stats()
        self._L.cedar_enc_get_stats(self._h, ctypes.byref(s))
        return {"frames": s.frames, "encode_wall_ms": s.encode_wall_ms, "wf_split": s.wf_split,
                "deblock_planes_split": bool(s.deblock_planes_split), "deblock_rows": bool(s.deblock_rows), "intra_rows": bool(s.intra_rows), "src_direct": bool(s.src_direct), "entropy_fused": bool(s.entropy_fused),
                "frames_pipelined": bool(s.frames_pipelined), "idr_lookahead": bool(s.idr_lookahead),
                "depth": s.depth, "deblock_streams": s.deblock_streams, "oversize_frames": s.oversize_frames, "frame_qp": s.frame_qp, "requantised_frames": s.requantised_frames, "scene_cuts": s.scene_cuts,
                "ms": {n: s.ms[i] for i, n in enumerate(STAGES)},
                "launches": {n: s.launches[i] for i, n in enumerate(STAGES)}}

    def reset_stats(self):
        self._L.cedar_enc_reset_stats(self._h)

    def debug_read(self, what, stream=0):
        n_mb = self.mb_w * self.mb_h
        shapes = {
            "mbinfo": (np.dtype([("type", "i1"), ("i16_mode", "i1"), ("chroma_mode", "i1"), ("cbp", "i1"),
                                 ("mvx", "<i2"), ("mvy", "<i2")]), (n_mb,)),
            "i4modes": (np.dtype("i1"), (n_mb, 16)),
            "nnz": (np.dtype("u1"), (n_mb, 24)),
            "coef": (np.dtype("<i2"), (n_mb, 408)),
            "recon_y": (np.dtype("u1"), (self.ch, self.cw)),
            "recon_uv": (np.dtype("u1"), (self.ch // 2, self.cw)),
            "src_y": (np.dtype("u1"), (self.ch, self.cw)),
            "src_uv": (np.dtype("u1"), (self.ch // 2, self.cw)),
            "mb_bits": (np.dtype("<u4"), (n_mb,)),
            "mb_qp": (np.dtype("u1"), (n_mb,)),
        }
        dt, shape = shapes[what]
        out = np.zeros(shape, dt)
        self._check(self._L.cedar_enc_debug_read(self._h, DBG[what], stream, out.ctypes.data, out.nbytes),
                    "debug_read")
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._L.cedar_enc_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
