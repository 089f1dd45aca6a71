"""Builds the in-tree native artefacts with hipcc for gfx950 (no JIT cache):

    cedarx_h264_encoder_amd/lib/libcedar_h264enc.so   the C-ABI library
    cedarx_h264_encoder_amd/lib/h264enc               the reference-compatible CLI

hipcc cross-compiles gfx950 without a GPU, so this also runs in the build
container. Objects are rebuilt only when a source or header is newer.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "lib")
OBJ = os.path.join(LIB, "obj")
ARCH = "gfx950"

SOURCES = ["enc_session.cpp", "enc_config.cpp", "enc_headers.cpp", "enc_submit.cpp", "enc_k_parallel.hip", "enc_k_wavefront.hip", "enc_k_deblock.hip", "enc_k_intra.hip", "enc_k_entropy.hip"]
LIB_PATH = os.path.join(LIB, "libcedar_h264enc.so")
CLI_PATH = os.path.join(LIB, "h264enc")


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(ROOT, "include", "cedar_h264enc.h"))
    return hs


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdr_time = _newest(_headers())
    flags = ["-O3", "-fPIC", "-std=c++17", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
             "-I" + os.path.join(ROOT, "include")]
    jobs = []
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src + ".o")
        objs.append(op)
        if force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr_time):
            cmd = [hipcc] + flags + ["-x", "hip", "-c", sp, "-o", op]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("build failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or force or not os.path.exists(LIB_PATH):
        run([hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB_PATH] + objs)
    cli_src = os.path.join(CSRC, "h264enc_cli.c")
    if os.path.exists(cli_src) and (force or not os.path.exists(CLI_PATH) or
                                    os.path.getmtime(CLI_PATH) < max(os.path.getmtime(cli_src), os.path.getmtime(LIB_PATH))):
        run(["gcc", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), cli_src, "-o", CLI_PATH,
             "-L" + LIB, "-lcedar_h264enc", "-Wl,-rpath,$ORIGIN"])
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
