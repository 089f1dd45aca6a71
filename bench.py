#!/usr/bin/env python3
"""bench.py -- macroblocks/s of the HIP encode path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: one frame of every stream
this rank owns (k_input -> intra or inter -> deblock || CAVLC..escape -> bytes on
the host). Workload = BASELINE.json configs[2] per GPU (1920x1080 nv12, I+P,
keyint 25, QP 24, synthetic pattern-B); with N GPUs every rank encodes its own
streams (configs[4]: independent streams shard with no collective), so scaling is
weak and `value` is total macroblocks of all ranks / max-over-ranks time.

Inputs are resident in HBM before the timed region (device-input entry point of
the C-ABI); the Annex-B bytes land in pinned host memory inside it.

Extra objects on the JSON line:
  roofline     dominant kernel: algorithmic bytes per launch / its average
               launch duration measured with HIP events on the kernel's stream
  cpu_baseline the scalar C oracle (oracle/, kind "port") timed on this box's
               host cores on a bounded sample of the same workload (rank 0, N=1)
"""
import argparse
import ctypes
import hashlib
import json
import os

# the encoder's frame pipelining wants more hardware queues than the HIP runtime's default 4;
# read when the runtime starts, so it has to be in the environment before the first HIP call
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# algorithmic HBM bytes per macroblock per kernel (DESIGN.md "Kernels"): each byte that
# must cross HBM once. 384 B = one macroblock of nv12.
KERNEL_BYTES_PER_MB = {
    "input": 384 + 384,          # read packed frame, write padded planes
    "intra_decide": 384,         # read source
    "intra_recon": 384 + 384,    # read source, write reconstruction
    "inter": 384 + 384 + 384,    # source + reference (each sample once) + reconstruction
    "deblock": 384 + 384,        # read + write reconstruction in place
    "entropy": 816 + 24 + 8,     # levels + counts + record read once (bitstream added separately)
}
PIPELINE_BYTES_PER_MB = {"I": 768.0, "P": 1152.0}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=25)
    ap.add_argument("--streams", type=int, default=1, help="lock-step streams per GPU")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--qp", type=int, default=24)
    ap.add_argument("--keyint", type=int, default=25)
    ap.add_argument("--frames", type=int, default=50, help="distinct synthetic frames per stream (cycled)")
    ap.add_argument("--cpu-frames", type=int, default=6, help="frames of the CPU baseline sample (0 = skip)")
    ap.add_argument("--batched-streams", type=str, default="8,32",
                    help="also report throughput with these many lock-step streams per GPU (0 = skip)")
    return ap.parse_args()


def run_rank(args, rank, world, local_rank):
    import numpy as np
    import torch
    import torch.distributed as dist
    import cedarx_h264_encoder_amd as ce

    # BENCH_FORCE_DEVICE lets several ranks share one GPU (rehearsal of the N > 1 path on a
    # one-GPU box, together with BENCH_DIST_BACKEND=gloo); the driver's runs never set it
    if os.environ.get("BENCH_FORCE_DEVICE") is not None:
        local_rank = int(os.environ["BENCH_FORCE_DEVICE"])
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    rdev = dev if (world == 1 or dist.get_backend() == "nccl") else torch.device("cpu")
    W, H = args.width, args.height
    mbs_per_frame = ((W + 15) // 16) * ((H + 15) // 16)

    def make_inputs(S):
        ins = []
        for s in range(S):
            gs = rank * S + s  # global stream id: seed offset s<<8
            fr = []
            for f in range(args.frames):
                y, uv = ce.gen_pattern("B", W, H, f, gs << 8)
                fr.append((torch.from_numpy(y).to(dev), torch.from_numpy(uv).to(dev)))
            ins.append(fr)
        torch.cuda.synchronize()
        return ins

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def prime(S, inputs):
        # Process warm-up, not a benchmark step: the HIP runtime grows its per-queue signal and
        # kernel-argument pools the first time many frames are in flight on a set of streams,
        # which stalls single submits by milliseconds. A throw-away session takes those stalls so
        # that the W warm-up frames of the measured session need not be many (the ring alone is 16
        # deep). Its output is discarded; nothing of it is reused by the measured session.
        with ce.CedarEncoder(width=W, height=H, qp=args.qp, keyframe_interval=args.keyint,
                             n_streams=S, device=local_rank) as pe:
            depth, q = pe.pipeline_depth, 0
            for i in range(3 * depth + args.keyint):
                if q == depth:
                    pe.collect()
                    q -= 1
                for s in range(S):
                    y, uv = inputs[s % len(inputs)][i % args.frames]
                    pe.set_device_input(s, y.data_ptr(), uv.data_ptr())
                pe.submit()
                q += 1
            while q:
                pe.collect()
                q -= 1

    def measure(S, steps, warmup, inputs, profile):
        prime(S, inputs)
        enc = ce.CedarEncoder(width=W, height=H, qp=args.qp, keyframe_interval=args.keyint,
                              n_streams=S, device=local_rank)
        nbytes = 0
        first_aus = []

        def feed(i):
            for s in range(S):
                y, uv = inputs[s % len(inputs)][i % args.frames]
                enc.set_device_input(s, y.data_ptr(), uv.data_ptr())
            enc.submit()

        # warmup (also keeps the first access units for the parity check). Every stage
        # is timed here to find the dominant one; the timed region below only times that stage:
        # events are stream markers and a dozen per frame cost ~10 % of a pipelined stream.
        enc.set_profiling(profile)
        depth = enc.pipeline_depth
        in_flight = 0
        # (pipelined like the timed region: the runtime grows its per-queue pools the first time
        # several frames are in flight, which stalls a few submits by milliseconds)
        for i in range(warmup):
            if in_flight == depth:
                enc.collect()
                if len(first_aus) < args.cpu_frames:
                    first_aus.append(enc.output(0))
                in_flight -= 1
            feed(i)
            in_flight += 1
        while in_flight:
            enc.collect()
            if len(first_aus) < args.cpu_frames:
                first_aus.append(enc.output(0))
            in_flight -= 1
        st_warm = enc.stats() if profile else None
        dominant = None
        if profile and st_warm and any(st_warm["ms"].values()):
            dominant = max(st_warm["ms"], key=lambda k: st_warm["ms"][k])
            # (every 4th frame: a stream marker behind a resident kernel costs that stream ~20 us)
            enc.set_profiling(True, stages=[dominant], every=4)
        else:
            enc.set_profiling(False)
        enc.reset_stats()
        barrier()
        t0 = time.perf_counter()
        # pipelined: as many frames in flight as the session allows (cedar_enc_submit / collect)
        depth = enc.pipeline_depth
        in_flight = 0
        for i in range(warmup, warmup + steps):
            if in_flight == depth:
                nbytes += enc.collect()
                in_flight -= 1
            feed(i)
            in_flight += 1
        while in_flight:
            nbytes += enc.collect()
            in_flight -= 1
        barrier()
        dt = time.perf_counter() - t0
        st = enc.stats()
        st["warmup"] = st_warm
        st["dominant"] = dominant
        enc.close()
        return dt, nbytes, st, first_aus

    S = args.streams
    sweep = [int(x) for x in str(args.batched_streams).split(",") if x.strip() and int(x) > 0 and int(x) != S]
    # at most 8 distinct synthetic streams are generated; larger batches reuse them cyclically
    inputs = make_inputs(max(S, min(8, max(sweep + [1]))))
    dt, nbytes, st, first_aus = measure(S, args.steps, args.warmup, inputs, os.environ.get("BENCH_NO_PROFILE") is None)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_max = float(t.item())
        b = torch.tensor([float(nbytes)], dtype=torch.float64, device=rdev)
        dist.all_reduce(b, op=dist.ReduceOp.SUM)
        nbytes_all = float(b.item())
    else:
        dt_max, nbytes_all = dt, float(nbytes)

    total_mbs = world * S * mbs_per_frame * args.steps
    value = total_mbs / dt_max

    batched = []
    for SB in sweep:
        bsteps = max(25, min(args.steps, 50))
        bdt, _, _, _ = measure(SB, bsteps, args.keyint, inputs, False)
        if world > 1:
            t = torch.tensor([bdt], dtype=torch.float64, device=rdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            bdt = float(t.item())
        v = world * SB * mbs_per_frame * bsteps / bdt
        batched.append({"streams_per_gpu": SB, "steps": bsteps, "value": v, "ms_per_step": bdt / bsteps * 1e3,
                        "pipeline_frac_of_hbm_peak": v * 1136.64 / 1e9 / HBM_PEAK_GBS})

    if rank != 0:
        return None

    # ---- roofline of the dominant kernel (HIP-event durations from the timed region) ----
    stage = st.get("dominant") or max(st["ms"], key=lambda k: st["ms"][k])
    launches = max(1, st["launches"][stage])
    avg_ms = max(st["ms"][stage] / launches, 1e-9)
    mb_per_launch = S * mbs_per_frame
    kname, tkey, per_mb = "k_" + stage, stage, KERNEL_BYTES_PER_MB[stage]
    if stage == "deblock":
        # one launch filters both planes (luma and chroma workgroups, or both in every wave pair)
        kname = "k_deblock2_planes"
    alg_bytes = per_mb * mb_per_launch
    if stage == "entropy":
        alg_bytes += nbytes / max(1, args.steps)
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            key = "%s@%dx%dx%d" % (tkey, W, H, S)
            traffic = tj.get(key)
        except Exception:  # noqa: BLE001
            traffic = None
    n_i = sum(1 for i in range(args.warmup, args.warmup + args.steps) if i % args.keyint == 0)
    pipe_bytes = (n_i * PIPELINE_BYTES_PER_MB["I"] + (args.steps - n_i) * PIPELINE_BYTES_PER_MB["P"]) \
        * S * mbs_per_frame * world + nbytes_all
    roofline = {
        "bound": "hbm", "kernel": kname, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
        "wf_split": st.get("wf_split"),
        "schedule": {k: st.get(k) for k in ("frames_pipelined", "idr_lookahead", "depth", "deblock_streams")},
        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
        "avg_launch_ms": round(avg_ms, 4), "launches": launches,
        "algorithmic_bytes_per_launch": int(alg_bytes),
        "pipeline_achieved_GBs": round(pipe_bytes / dt_max / 1e9, 3),
        "pipeline_frac": pipe_bytes / dt_max / 1e9 / HBM_PEAK_GBS,
        # every stage, from the warm-up frames; the dominant stage above is timed inside the timed region
        "stage_ms_per_launch_warmup": ({k: round(st["warmup"]["ms"][k] / max(1, st["warmup"]["launches"][k]), 4)
                                        for k in st["warmup"]["ms"]} if st.get("warmup") else None),
    }

    # ---- CPU baseline: the scalar C oracle on a bounded sample (rank 0, N=1 only) ----
    cpu = None
    bitexact = None
    if world == 1 and args.cpu_frames > 0:
        from tests import oracle_lib as ol
        oe = ol.OracleEncoder(ol.default_config(W, H, qp=args.qp, keyint=args.keyint))
        frames = [ce.gen_pattern("B", W, H, f, 0) for f in range(args.cpu_frames)]
        t0 = time.perf_counter()
        cpu_aus = [oe.encode(y, uv) for y, uv in frames]
        cdt = time.perf_counter() - t0
        n_cmp = min(len(cpu_aus), len(first_aus))
        bitexact = n_cmp > 0 and all(cpu_aus[i] == first_aus[i] for i in range(n_cmp))
        cpu = {"value": args.cpu_frames * mbs_per_frame / cdt, "unit": "macroblocks/s", "cores": 1,
               "kind": "port",
               "sample": "%d frames %dx%d pattern-B (1 IDR + %d P), QP %d, single thread, gcc -O2" % (
                   args.cpu_frames, W, H, args.cpu_frames - 1, args.qp),
               "seconds": round(cdt, 2),
               "fnv_sha": hashlib.sha1(b"".join(cpu_aus)).hexdigest()[:16],
               "gpu_sha": hashlib.sha1(b"".join(first_aus[:n_cmp])).hexdigest()[:16]}
        # (ii) of SURVEY 8(d): the host cores of a one-GPU box, one independent stream per core: N
        # processes of the scalar command line (oracle/build/h264enc_scalar) on the same sample
        # (threads in this process would share one address space and contend on page faults)
        import subprocess
        import tempfile
        ncores = max(1, min(len(os.sched_getaffinity(0)), 16))
        nfr = args.cpu_frames  # the same sample, so the same IDR / P mix
        scalar = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle", "build", "h264enc_scalar")
        if not os.path.exists(scalar):
            subprocess.run(["make", "-s", "-C", os.path.dirname(os.path.dirname(scalar))], check=True)
        with tempfile.TemporaryDirectory() as td:
            src = os.path.join(td, "sample.nv12")
            with open(src, "wb") as fh:
                for y, uv in frames[:nfr]:
                    fh.write(y.tobytes())
                    fh.write(uv.tobytes())
            env = dict(os.environ, H264ENC_QP=str(args.qp), H264ENC_KEYINT=str(args.keyint))
            t0 = time.perf_counter()
            procs = [subprocess.Popen([scalar, src, str(W), str(H), os.path.join(td, "o%d.264" % i)],
                                      stdout=subprocess.DEVNULL, env=env) for i in range(ncores)]
            rcs = [pr.wait() for pr in procs]
            mdt = time.perf_counter() - t0
            with open(os.path.join(td, "o0.264"), "rb") as fh:
                same = fh.read() == b"".join(cpu_aus[:nfr])
        if any(rcs) or not same:
            raise SystemExit("cpu_baseline all-cores leg failed (rc %s, same bytes %s)" % (rcs, same))
        cpu["all_cores"] = {"value": ncores * nfr * mbs_per_frame / mdt, "cores": ncores,
                            "sample": "%d processes x %d frames of the same sample, one per core" % (ncores, nfr),
                            "seconds": round(mdt, 2)}

    out = {
        "metric": "macroblocks/s @ fixed QP, 1080p nv12",
        "value": value,
        "unit": "macroblocks/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt_max / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "%dx%d nv12 I+P keyint %d QP %d, pattern-B, %d stream(s) per GPU "
                               "(BASELINE configs[2]; N GPUs = independent streams, configs[4])" % (
                                   W, H, args.keyint, args.qp, S),
                   "streams_per_gpu": S, "macroblocks_per_frame": mbs_per_frame,
                   "parallelism": "streams x%d, no collective" % world,
                   "realtime_4k30_multiple": value / 972000.0,
                   "bitstream_bytes_per_step": nbytes_all / args.steps},
        "roofline": roofline,
        "cpu_baseline": cpu,
        "bitexact_vs_cpu": bitexact,
        "batched": batched,
    }
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        # launched without torchrun: spawn it ourselves so `python bench.py --gpus N` works
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    import torch  # noqa: F401  (first, so the encoder library shares torch's HIP runtime)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    try:
        out = run_rank(args, rank, world, local_rank)
        if rank == 0 and out is not None:
            print(json.dumps(out), flush=True)
    finally:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
