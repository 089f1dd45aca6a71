"""cedarx_h264_encoder_amd -- MI355X-native H.264 baseline encode path.

The product is the C-ABI shared library built from csrc/ (HIP kernels for
gfx950 + the host session code); this package is a thin ctypes binding over it
that mirrors the reference's session interface (kernel/cedar_ioctl.h:1-48,
userspace/h264enc.c:47-117): open -> config -> fill input buffers -> encode ->
read bytestream -> close. There is no CPU fallback: without the built library
or without a HIP device every entry point raises.
"""
from .encoder import (CedarEncoder, EncoderConfig, EncoderError, STAGES, gen_pattern, library_path,
                      load_library)

__all__ = ["CedarEncoder", "EncoderConfig", "EncoderError", "STAGES", "gen_pattern", "library_path",
           "load_library"]
