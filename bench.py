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
`--slices-across-gpus` is configs[3] instead: rank k encodes slice k of ONE picture
(3840x2160 by default), the host concatenates the slice NALs; strong scaling.
`--gops-across-gpus` shards ONE stream by closed GOPs: GOP g on rank
g mod N, a step = one GOP per rank, the byte ranges concatenate to the serial stream.

Inputs are resident in HBM before the timed region (device-input entry point of
the C-ABI); the Annex-B bytes land in pinned host memory inside it.

The timed region carries the GOP's I:P mix whatever K and W are: in front of the W
warm-up frames the measured session encodes a few alignment frames, so that the K
timed frames hold ceil(K / keyint) IDR pictures, the first of them min(ring depth,
(K-1) mod keyint) frames in -- where it sits in steady state, behind a ring of P
pictures (`config.timed_region`; BENCH_IDR_FIRST=1 puts it at frame 0 instead).

Extra objects on the JSON line:
  roofline     dominant kernel: algorithmic bytes per launch / its average launch
               duration (HIP events on the kernel's stream). `avg_launch_ms` is measured
               in the timed region, where a pipelined deblock kernel is RESIDENT and
               polls for its frame's analysis; `work_ms` is the same kernel in a session
               whose frames run back to back (duration = work) and is what `achieved`
               divides by. `true_bound` names what really limits the kernel (the SAD
               issue rate for k_inter, the dependency chain for the wavefront kernels)
               and the fraction of THAT ceiling; `kernels` lists every stage the same way.
  cpu_baseline the scalar C oracle (oracle/, kind "port") timed on this box's
               host cores on a bounded sample of the same workload (rank 0, N=1)
  region_runs  the spread of `value`: --regions (5) consecutive timed regions of --steps frames of the measured session,
               each with its IDR pictures where the first had them and bracketed like it; `value` is the first
  gop_parallel the same ONE stream with G closed GOPs encoded side by side
               (cedar_enc_config_ext.gop_parallel; same bytes, G*keyint frames of latency)
  batched      S lock-step streams per GPU
  configs      BASELINE configs[1] (854x480 QP 30) and configs[3] in its one-GPU forms (3840x2160 QP 24, one slice and
               4 slices), 25 timed steps each (--other-configs 0 skips them)
"""
import argparse
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
KERNEL_NAME = {"input": "k_input", "intra_decide": "k_intra_decide", "intra_recon": "k_intra_recon2", "inter": "k_inter",
               "deblock": "k_deblock2_planes", "entropy": "k_cavlc2..k_copy_out"}
KERNEL_NAME_FUSED_ENTROPY = "k_cavlc2, k_scan, k_pack_out"
PIPELINE_BYTES_PER_MB = {"I": 768.0, "P": 1152.0}
# What really bounds each kernel. k_inter: the exhaustive +-16 search is 33*33
# candidates x 256 absolute differences per macroblock; every byte-SAD instruction of gfx950 retires
# ~1 absolute difference per lane and cycle (tools/dev/sad_bench.hip: v_sad_u8 4 per 4.17 cycles,
# v_qsad_pk_u16_u8 16 per 14.8), so the chip's ceiling is 256 CUs x 4 SIMDs x 64 lanes x 2.4 GHz.
SAD_OPS_PER_MB = 33 * 33 * 256
SAD_PEAK_OPS = 256 * 4 * 64 * 2.4e9 * (4 / 4.17)
# Round 3: the search is still exhaustive in its result but no longer in its work (successive elimination on 8x8 box
# sums), so byte-SADs per second say nothing any more. What bounds the kernel is plain VALU issue: one
# wave instruction per SIMD every 4 cycles, and k_inter<0,0,0> issues ~1210 of them per macroblock on its common path
# (one pass of exact SADs over the survivors; tools/dev/isa_buckets.py on the gfx950 listing, a quarter-rate
# v_qsad_pk_u16_u8 counted as 4).
INTER_VALU_SLOTS_PER_MB = 1210
VALU_PEAK_SLOTS = 256 * 4 * 2.4e9 / 4


def valu_counters():
    """profiles/valu.json (tools/collect_r04_valu.sh + tools/pmc_valu_summary.py): SQ counters per kernel of the w1 leg (one
    1080p stream, frames back to back) and the s32 leg (32 lock-step streams). An issue slot = one SQ_ACTIVE_INST_VALU
    count = 4 SIMD cycles with a VALU instruction in flight. Static: counters of a committed profile, not of this run."""
    try:
        v = json.load(open(os.path.join(ROOT, "profiles", "valu.json")))
    except Exception:  # noqa: BLE001
        return None
    tag = sorted(k for k in v if k.endswith("_s32"))
    tag1 = sorted(k for k in v if k.endswith("_w1"))
    if not tag or not tag1:
        return None
    s32, w1 = v[tag[-1]], v[tag1[-1]]
    idr_only = ("k_intra", "k_deblock_rows<1>", "k_deblock_rows<5>", "k_deblock2_planes<1", "k_deblock2_planes<5", "k_probe", "k_flag")

    def per_mb(leg):
        return {k: round(d["active_slots_per_macroblock"], 1) for k, d in leg.items()
                if d.get("active_slots_per_macroblock", 0) >= 0.05 and not k.startswith("k_probe")}

    def p_picture(leg):
        return sum(d.get("active_slots_per_macroblock", 0.0) for k, d in leg.items() if not k.startswith(idr_only))

    inter = next((d for k, d in s32.items() if k.startswith("k_inter<0, false, false>")), None)
    return {"source": "profiles/valu.json: rocprofv3 --pmc SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_WAVES / SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE, "
                      "separate passes, legs %s and %s (static: a committed profile, not this run)" % (tag1[-1], tag[-1]),
            "slot": "4 SIMD cycles with a VALU instruction in flight (one SQ_ACTIVE_INST_VALU count)",
            "slots_per_macroblock_one_stream": per_mb(w1), "slots_per_macroblock_32_streams": per_mb(s32),
            "p_picture_slots_per_macroblock_one_stream": round(p_picture(w1), 1),
            "p_picture_slots_per_macroblock_32_streams": round(p_picture(s32), 1),
            "k_inter": None if inter is None else {
                "instructions_per_macroblock": round(inter["valu_per_macroblock"], 1),
                "slots_per_macroblock": round(inter["active_slots_per_macroblock"], 1),
                "static_isa_count": INTER_VALU_SLOTS_PER_MB,
                "slot_share_while_it_runs_alone": round(inter["active_slot_share"], 4)}}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=25)
    ap.add_argument("--streams", type=int, default=1, help="lock-step streams per GPU")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--qp", type=int, default=24)
    ap.add_argument("--keyint", type=int, default=25)
    ap.add_argument("--frames", type=int, default=50, help="distinct synthetic frames per stream (cycled)")
    ap.add_argument("--cpu-frames", type=int, default=6, help="frames of the CPU baseline sample (0 = skip)")
    ap.add_argument("--batched-streams", type=str, default="8,32",
                    help="also report throughput with these many lock-step streams per GPU (0 = skip)")
    ap.add_argument("--gop-parallel", type=str, default="8",
                    help="also report ONE stream with this many closed GOPs side by side (0 = skip)")
    ap.add_argument("--other-configs", type=int, default=1,
                    help="1: also time BASELINE configs[1] (854x480 QP 30) and configs[3] in its one-GPU forms "
                         "(3840x2160 QP 24 as 1 slice and as 4 slices) for 25 steps each (N=1 only); 0 = skip")
    ap.add_argument("--regions", type=int, default=5,
                    help="timed regions of --steps frames of the measured session: `value` is the first, `region_runs` lists all")
    ap.add_argument("--slices-across-gpus", action="store_true",
                    help="configs[3]: rank k encodes slice k of one picture (default 3840x2160), no collective")
    ap.add_argument("--gops-across-gpus", action="store_true",
                    help="ONE stream, GOP g on rank g mod N (cedar_enc_set_frame_index), no collective; a step = one GOP per rank")
    a = ap.parse_args()
    if not a.width:
        a.width, a.height = (3840, 2160) if a.slices_across_gpus else (1920, 1080)
    return a


def run_rank(args, rank, world, local_rank):
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
    K = args.keyint
    mbs_per_frame = ((W + 15) // 16) * ((H + 15) // 16)

    # what the sessions of this run encode: the headline's geometry, or (--other-configs) another BASELINE config's
    cur = {"W": W, "H": H, "qp": args.qp, "slices": 0}

    def make_inputs(S, nframes=None):
        ins = []
        for s in range(S):
            # global stream id: seed offset s<<8; one picture / one stream sharded: the same on every rank
            gs = (0 if (args.slices_across_gpus or args.gops_across_gpus) else rank * S) + s
            fr = []
            for f in range(nframes or args.frames):
                y, uv = ce.gen_pattern("B", cur["W"], cur["H"], f, gs << 8)
                fr.append((torch.from_numpy(y).to(dev), torch.from_numpy(uv).to(dev)))
            ins.append(fr)
        torch.cuda.synchronize()
        return ins

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def allmax(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def allsum(x):
        if world == 1:
            return x
        t = torch.tensor([float(x)], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def session(S, **kw):
        if args.slices_across_gpus:   # this rank's slice of the one picture
            kw.update(slice_count=world, slice_first=rank, slice_num=1)
        elif cur["slices"]:
            kw.update(slice_count=cur["slices"])
        return ce.CedarEncoder(width=cur["W"], height=cur["H"], qp=cur["qp"], keyframe_interval=K, n_streams=S, device=local_rank, **kw)

    class Pump:
        """frames through submit / collect with the ring as full as the session allows"""

        def __init__(self, enc, S, inputs, keep=None, keep_n=0):
            self.enc, self.S, self.inputs, self.keep, self.keep_n = enc, S, inputs, keep, keep_n
            self.depth, self.in_flight, self.nbytes = enc.pipeline_depth, 0, 0

        def take(self):
            self.nbytes += self.enc.collect()
            self.in_flight -= 1
            if self.keep is not None and len(self.keep) < self.keep_n:
                self.keep.append(self.enc.output(0))

        def feed(self, first, n):
            for i in range(first, first + n):
                if self.in_flight == self.depth:
                    self.take()
                for s in range(self.S):
                    y, uv = self.inputs[s % len(self.inputs)][i % len(self.inputs[0])]
                    self.enc.set_device_input(s, y.data_ptr(), uv.data_ptr())
                self.enc.submit()
                self.in_flight += 1
            return self

        def drain(self):
            while self.in_flight:
                self.take()
            return self

    def prime(S, inputs):
        # Process warm-up, not a benchmark step: the HIP runtime grows its per-queue signal and
        # kernel-argument pools the first time many frames are in flight on a set of streams,
        # which stalls single submits by milliseconds. A throw-away session takes those stalls so
        # that the W warm-up frames of the measured session need not be many (the ring alone is 16
        # deep). Its output is discarded; nothing of it is reused by the measured session.
        with session(S) as pe:
            Pump(pe, S, inputs).feed(0, 3 * pe.pipeline_depth + K).drain()

    def measure(S, steps, warmup, inputs, profile, keep_n=0, regions=1):
        prime(S, inputs)
        enc = session(S)
        first_aus = []
        # alignment frames + warm-up (also keeps the first access units for the parity check). Every
        # stage is timed here to find the dominant one; the timed region below only times that stage:
        # events are stream markers and a dozen per frame cost ~10 % of a pipelined stream.
        # Where the IDR pictures fall in the timed region: ceil(steps / keyint) of them, the first one
        # min(depth, ...) frames in -- the steady state, where an IDR picture is submitted behind a ring
        # of P pictures and the look-ahead runs it beside them (BENCH_IDR_FIRST=1: at frame 0 instead,
        # on the empty pipeline the barrier leaves -- the worst case, its 2.6 ms fully exposed)
        first_idr = 0 if os.environ.get("BENCH_IDR_FIRST") else min(enc.pipeline_depth, steps - 1, (steps - 1) % K)
        pad = (-(warmup + first_idr)) % K
        while pad + warmup < keep_n:
            pad += K
        enc.set_profiling(profile)
        Pump(enc, S, inputs, first_aus, keep_n).feed(0, pad + warmup).drain()
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
        pm = Pump(enc, S, inputs).feed(pad + warmup, steps).drain()
        barrier()
        dt = time.perf_counter() - t0
        st = enc.stats()
        # the spread of the number: `regions - 1` more timed regions of the same session, each of `steps` frames with its IDR
        # pictures where the first region had them (untimed frames in between bring the GOP phase back), each bracketed
        # like the first. `value` stays the first region's.
        st["region_seconds"] = [dt]
        nxt = pad + warmup + steps
        for _ in range(max(0, regions - 1)):
            gap = (-steps) % K
            Pump(enc, S, inputs).feed(nxt, gap).drain()
            nxt += gap
            barrier()
            t1 = time.perf_counter()
            Pump(enc, S, inputs).feed(nxt, steps).drain()
            barrier()
            st["region_seconds"].append(time.perf_counter() - t1)
            nxt += steps
        st["warmup"] = st_warm
        st["dominant"] = dominant
        st["first_frame"] = pad + warmup
        enc.close()
        return dt, pm.nbytes, st, first_aus

    def measure_work(S, inputs):
        """per-stage kernel time with the frames of the session back to back: durations are work"""
        os.environ["CEDAR_ROW_PIPE"] = "0"
        try:
            with session(S) as enc:
                enc.set_profiling(True)
                for i in range(K + 8):
                    for s in range(S):
                        y, uv = inputs[s % len(inputs)][i % args.frames]
                        enc.set_device_input(s, y.data_ptr(), uv.data_ptr())
                    enc.encode()
                st = enc.stats()
        finally:
            del os.environ["CEDAR_ROW_PIPE"]
        return {k: st["ms"][k] / max(1, st["launches"][k]) for k in st["ms"]}

    def measure_gop_parallel(G, inputs, steps):
        """ONE stream, G closed GOPs side by side: frames in and out in stream order, whole groups"""
        n = max(1, -(-steps // (G * K))) * G * K
        prime(G, inputs)
        with session(1, gop_parallel=G) as enc:
            barrier()
            t0 = time.perf_counter()
            pm = Pump(enc, 1, inputs).feed(0, n)
            enc.flush()
            pm.drain()
            barrier()
            dt = time.perf_counter() - t0
        return n, dt

    if args.gops_across_gpus:
        return run_gop_shards(args, rank, world, local_rank, ce, torch, dist, make_inputs, barrier, allmax, mbs_per_frame)
    S = args.streams
    sweep = [int(x) for x in str(args.batched_streams).split(",") if x.strip() and int(x) > 0 and int(x) != S]
    gsweep = [int(x) for x in str(args.gop_parallel).split(",") if x.strip() and int(x) > 1]
    if args.slices_across_gpus:
        sweep, gsweep = [], []
    # at most 8 distinct synthetic streams are generated; larger batches reuse them cyclically
    inputs = make_inputs(max(S, min(8, max(sweep + [1]))))
    profile = os.environ.get("BENCH_NO_PROFILE") is None
    keep_n = args.cpu_frames if world == 1 else (3 if args.slices_across_gpus else 0)
    dt, nbytes, st, first_aus = measure(S, args.steps, args.warmup, inputs, profile, keep_n=keep_n, regions=args.regions)
    dt_max = allmax(dt)
    region_dts = [allmax(x) for x in st["region_seconds"]]
    nbytes_all = allsum(nbytes)
    slices_equal = None
    if args.slices_across_gpus and world > 1:
        # verification, outside the timed region (the data path itself has no collective): the slice
        # NALs of the first pictures from every rank, concatenated in slice order by rank 0, must be
        # the access units of ONE session coding all the slices
        parts = [None] * world
        dist.all_gather_object(parts, first_aus)
        if rank == 0:
            with ce.CedarEncoder(width=W, height=H, qp=args.qp, keyframe_interval=K, n_streams=S, device=local_rank,
                                 slice_count=world) as one:
                whole = []
                for i in range(len(first_aus)):
                    for s in range(S):
                        y, uv = inputs[s % len(inputs)][i % args.frames]
                        one.set_device_input(s, y.data_ptr(), uv.data_ptr())
                    one.encode()
                    whole.append(one.output(0))
            joined = [b"".join(parts[r][i] for r in range(world)) for i in range(len(first_aus))]
            slices_equal = joined == whole
            if not slices_equal:
                raise SystemExit("slices from %d ranks do not concatenate to the one-session stream" % world)

    if args.slices_across_gpus:   # one picture per step, whatever the rank count
        total_mbs = S * mbs_per_frame * args.steps
    else:
        total_mbs = world * S * mbs_per_frame * args.steps
    value = total_mbs / dt_max

    batched = []
    for SB in sweep:
        bsteps = max(25, min(args.steps, 50))
        bdt, _, _, _ = measure(SB, bsteps, K, inputs, False)
        bdt = allmax(bdt)
        v = world * SB * mbs_per_frame * bsteps / bdt
        batched.append({"streams_per_gpu": SB, "steps": bsteps, "value": v, "ms_per_step": bdt / bsteps * 1e3,
                        "pipeline_frac_of_hbm_peak": v * 1136.64 / 1e9 / HBM_PEAK_GBS})
    gop_par = []
    for G in gsweep:
        n, gdt = measure_gop_parallel(G, inputs, args.steps)
        gdt = allmax(gdt)
        v = world * mbs_per_frame * n / gdt
        gop_par.append({"gops_side_by_side": G, "frames": n, "value": v, "ms_per_frame": gdt / n * 1e3,
                        "latency_frames": 2 * (G - 1) * K + 9,   # gop_low_latency = 0: the next group's IDR step hides
                        "pipeline_frac_of_hbm_peak": v * 1136.64 / 1e9 / HBM_PEAK_GBS})
    work_ms = measure_work(S, inputs) if profile else None

    # BASELINE configs[1] and configs[3] (in its one-GPU forms) on the same line: 25 timed steps each, like `batched`
    others = []
    if args.other_configs and world == 1 and not args.slices_across_gpus and (W, H, S) == (1920, 1080, 1):
        for name, ow, oh, oqp, osl in (("configs[1]: 854x480 nv12 I+P QP 30", 854, 480, 30, 0),
                                      ("configs[3] on one GPU: 3840x2160 nv12 I+P QP 24, one slice", 3840, 2160, 24, 0),
                                      ("configs[3] on one GPU: 3840x2160 nv12 I+P QP 24, 4 slices (34/34/34/33 rows)", 3840, 2160, 24, 4)):
            cur.update(W=ow, H=oh, qp=oqp, slices=osl)
            oin = make_inputs(1, K + 1)
            osteps = 25
            odt, onb, _, _ = measure(1, osteps, K, oin, False)
            ombs = ((ow + 15) // 16) * ((oh + 15) // 16)
            ov = ombs * osteps / odt
            others.append({"workload": name + ", keyint %d, pattern-B, 1 stream" % K, "steps": osteps, "value": ov,
                           "ms_per_step": odt / osteps * 1e3, "macroblocks_per_frame": ombs,
                           "realtime_30fps_multiple": ov / (ombs * 30.0), "bitstream_bytes_per_step": onb / osteps,
                           "pipeline_frac_of_hbm_peak": ov * 1136.64 / 1e9 / HBM_PEAK_GBS})
            del oin
        cur.update(W=W, H=H, qp=args.qp, slices=0)

    if rank != 0:
        return None

    # ---- roofline of the dominant kernel ----
    stage = st.get("dominant") or max(st["ms"], key=lambda k: st["ms"][k])
    launches = max(1, st["launches"][stage])
    avg_ms = max(st["ms"][stage] / launches, 1e-9)   # in the timed region: resident time for a pipelined deblock kernel
    my_mbs = mbs_per_frame // world if args.slices_across_gpus else mbs_per_frame
    mb_per_launch = S * my_mbs
    wms = (work_ms or {}).get(stage) or avg_ms

    def kernel_line(k, ms):
        alg = KERNEL_BYTES_PER_MB[k] * mb_per_launch + (nbytes / max(1, args.steps) if k == "entropy" else 0)
        ach = alg / (ms * 1e-3) / 1e9
        name = ("k_deblock_rows" if k == "deblock" and st.get("deblock_rows") else "k_intra_rows" if k == "intra_recon" and st.get("intra_rows")
                else KERNEL_NAME_FUSED_ENTROPY if k == "entropy" and st.get("entropy_fused") else KERNEL_NAME[k])
        line = {"kernel": name, "work_ms": round(ms, 4), "algorithmic_bytes_per_launch": int(alg),
                "hbm_GBs": round(ach, 2), "hbm_frac": ach / HBM_PEAK_GBS}
        if k == "inter":
            per_mb = ((valu or {}).get("k_inter") or {}).get("slots_per_macroblock") or INTER_VALU_SLOTS_PER_MB
            slots = per_mb * mb_per_launch / (ms * 1e-3)
            line["true_bound"] = {"kind": "valu issue (wave instructions; the search prunes by exact successive elimination)",
                                  "valu_slots_per_macroblock": per_mb,
                                  "valu_slots_source": "SQ_ACTIVE_INST_VALU (profiles/valu.json)" if valu else "static ISA count",
                                  "achieved_Gslots": round(slots / 1e9, 1), "peak_Gslots": round(VALU_PEAK_SLOTS / 1e9, 1),
                                  "frac": slots / VALU_PEAK_SLOTS,
                                  "exhaustive_equivalent_Tsad": round(SAD_OPS_PER_MB * mb_per_launch / (ms * 1e-3) / 1e12, 1)}
        elif k in ("deblock", "intra_recon"):
            mb_w, mb_h = (W + 15) // 16, (H + 15) // 16
            nsteps = mb_w + (mb_h - 1) * (1 if k == "deblock" else 2)
            if k == "intra_recon" and st.get("intra_rows"):   # steps are 4x4 blocks: T = X + 2 Y (enc_k_intra.hip)
                nsteps = 4 * mb_w + 2 * (4 * mb_h - 1)
            line["true_bound"] = {"kind": "dependency latency: anti-diagonal wavefront" + (" of 4x4 blocks" if k == "intra_recon" and st.get("intra_rows") else ""),
                                  "wavefront_steps": nsteps, "us_per_step": round(ms * 1e3 / nsteps, 3)}
            if k == "deblock" and st.get("deblock_rows"):   # a workgroup (CU) per band of 4 rows: rows above cross L2
                line["true_bound"]["band_boundaries"] = max(0, (mb_h + 3) // 4 - 1)
        else:
            line["true_bound"] = {"kind": "hbm (latency at this size)", "frac": ach / HBM_PEAK_GBS}
        return line

    valu = valu_counters()
    dom = kernel_line(stage, wms)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tkey = "deblock_rows" if stage == "deblock" and st.get("deblock_rows") else stage
            traffic = json.load(open(tpath)).get("%s@%dx%dx%d" % (tkey, W, H, S))
        except Exception:  # noqa: BLE001
            traffic = None
    first = st["first_frame"]
    n_i = sum(1 for i in range(first, first + args.steps) if i % K == 0)
    pipe_bytes = (n_i * PIPELINE_BYTES_PER_MB["I"] + (args.steps - n_i) * PIPELINE_BYTES_PER_MB["P"]) \
        * total_mbs / args.steps + nbytes_all
    roofline = {
        "bound": "hbm", "kernel": dom["kernel"], "achieved": dom["hbm_GBs"], "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": dom["hbm_GBs"] / HBM_PEAK_GBS, "traffic": traffic,
        # NOT a counter of this run: the HBM bytes per launch of this kernel from the committed rocprofv3 --pmc passes
        # (FETCH_SIZE, WRITE_SIZE in separate runs, as the guide prescribes) of the same command
        "traffic_source": "profiles/traffic.json (static: rocprofv3 --pmc passes of an earlier run, see profiles/)" if traffic is not None else None,
        "work_ms": dom["work_ms"], "avg_launch_ms": round(avg_ms, 4), "launches": launches,
        "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
        "true_bound": dom["true_bound"],
        "wf_split": st.get("wf_split"),
        "schedule": {k: st.get(k) for k in ("frames_pipelined", "idr_lookahead", "depth", "deblock_streams", "deblock_rows", "intra_rows", "src_direct", "entropy_fused")},
        "pipeline_achieved_GBs": round(pipe_bytes / dt_max / 1e9, 3),
        "pipeline_frac": pipe_bytes / dt_max / 1e9 / HBM_PEAK_GBS,
        # every stage with frames back to back (duration = work), against its own bound
        "kernels": [kernel_line(k, v) for k, v in (work_ms or {}).items() if v > 0 and not (k == "input" and st.get("src_direct"))],
        # every stage as launched in the measured (pipelined) session, from its warm-up frames
        "stage_ms_per_launch_pipelined": ({k: round(st["warmup"]["ms"][k] / max(1, st["warmup"]["launches"][k]), 4)
                                           for k in st["warmup"]["ms"]} if st.get("warmup") else None),
        "valu": valu,
    }
    if valu:
        # what the lock-step batch of this run does with the chip's VALU issue: slots per macroblock of a P picture's kernels
        # (counters) x the macroblock rate measured here, against 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per slot
        b32 = next((b for b in batched if b["streams_per_gpu"] == 32), None)
        if b32:
            g = valu["p_picture_slots_per_macroblock_32_streams"] * b32["value"]
            valu["batch_32_streams"] = {"macroblocks_per_s": b32["value"], "achieved_Gslots": round(g / 1e9, 1),
                                        "peak_Gslots": round(VALU_PEAK_SLOTS / 1e9, 1), "frac": g / VALU_PEAK_SLOTS}
        g1 = valu["p_picture_slots_per_macroblock_one_stream"] * value
        valu["headline"] = {"macroblocks_per_s": value, "achieved_Gslots": round(g1 / 1e9, 1), "frac": g1 / VALU_PEAK_SLOTS}

    # ---- CPU baseline: the scalar C oracle on a bounded sample (rank 0, N=1 only) ----
    cpu = None
    bitexact = None
    if world == 1 and args.cpu_frames > 0:
        from tests import oracle_lib as ol
        oe = ol.OracleEncoder(ol.default_config(W, H, qp=args.qp, keyint=K))
        frames = [ce.gen_pattern("B", W, H, f, 0) for f in range(args.cpu_frames)]
        t0 = time.perf_counter()
        cpu_aus = [oe.encode(y, uv) for y, uv in frames]
        cdt = time.perf_counter() - t0
        n_cmp = min(len(cpu_aus), len(first_aus))
        bitexact = n_cmp == args.cpu_frames and all(cpu_aus[i] == first_aus[i] for i in range(n_cmp))
        cpu = {"value": args.cpu_frames * mbs_per_frame / cdt, "unit": "macroblocks/s", "cores": 1,
               "kind": "port",
               "sample": "%d frames %dx%d pattern-B (1 IDR + %d P), QP %d, single thread, gcc -O2" % (
                   args.cpu_frames, W, H, args.cpu_frames - 1, args.qp),
               "seconds": round(cdt, 2),
               # the SAME access units on both sides: the first cpu_frames of stream 0
               "access_units_compared": n_cmp,
               "sha1_cpu": hashlib.sha1(b"".join(cpu_aus[:n_cmp])).hexdigest()[:16],
               "sha1_gpu": hashlib.sha1(b"".join(first_aus[:n_cmp])).hexdigest()[:16]}
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
            env = dict(os.environ, H264ENC_QP=str(args.qp), H264ENC_KEYINT=str(K))
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

    if args.slices_across_gpus:
        workload = ("%dx%d nv12 I+P keyint %d QP %d, pattern-B, ONE picture in %d slices, slice k on GPU k "
                    "(BASELINE configs[3]; no collective, the host concatenates the slice NALs)" % (W, H, K, args.qp, world))
        par = "slices x%d of one picture, no collective" % world
    else:
        workload = ("%dx%d nv12 I+P keyint %d QP %d, pattern-B, %d stream(s) per GPU "
                    "(BASELINE configs[2]; N GPUs = independent streams, configs[4])" % (W, H, K, args.qp, S))
        par = "streams x%d, no collective" % world
    out = {
        "metric": "macroblocks/s @ fixed QP, 1080p nv12" if (W, H) == (1920, 1080) else "macroblocks/s @ fixed QP, %dx%d nv12" % (W, H),
        "value": value,
        "unit": "macroblocks/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt_max / args.steps * 1e3,
        # the same session's consecutive timed regions (same steps, same IDR placement), the first one is `value`
        "region_runs": [total_mbs / x for x in region_dts],
        "higher_is_better": True,
        "scaling": "strong" if args.slices_across_gpus else "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": workload,
                   "streams_per_gpu": S, "macroblocks_per_frame": mbs_per_frame,
                   "parallelism": par,
                   "timed_region": {"first_frame": first, "alignment_frames_before_warmup": first - args.warmup,
                                    "idr_pictures": n_i, "p_pictures": args.steps - n_i,
                                    "first_idr_at_step": (-first) % K},
                   "idr_in_timed_region": n_i,
                   "slices_concatenate_to_one_session_stream": slices_equal,
                   "realtime_4k30_multiple": value / 972000.0,
                   "bitstream_bytes_per_step": nbytes_all / args.steps},
        "roofline": roofline,
        "cpu_baseline": cpu,
        "bitexact_vs_cpu": bitexact,
        "gop_parallel": gop_par,
        "batched": batched,
        "configs": others,
    }
    return out


def run_gop_shards(args, rank, world, local_rank, ce, torch, dist, make_inputs, barrier, allmax, mbs_per_frame):
    """ONE stream sharded by closed GOPs: rank r codes GOPs r, r + N, ... (`cedar_enc_set_frame_index`
    moves its session on), the host would concatenate the byte ranges in GOP order. A step is one GOP per
    rank; the run gathers the first GOPs of every rank and checks the concatenation against ONE session
    coding the same frames serially (outside the timed region; the data path has no collective)."""
    W, H, K = args.width, args.height, args.keyint
    inputs = make_inputs(1)[0]
    nf = len(inputs)

    def gop(enc, g, keep=None):
        enc.set_frame_index(g * K)
        depth, q, nb = enc.pipeline_depth, 0, 0
        for t in range(K):
            if q == depth:
                nb += enc.collect()
                q -= 1
                if keep is not None:
                    keep.append(enc.output(0))
            y, uv = inputs[(g * K + t) % nf]
            enc.set_device_input(0, y.data_ptr(), uv.data_ptr())
            enc.submit()
            q += 1
        while q:
            nb += enc.collect()
            q -= 1
            if keep is not None:
                keep.append(enc.output(0))
        return nb

    enc = ce.CedarEncoder(width=W, height=H, qp=args.qp, keyframe_interval=K, device=local_rank)
    mine = []
    for k in range(max(1, args.warmup // K) + 1):          # untimed: this rank's first GOPs (kept for the check)
        gop(enc, k * world + rank, mine if k < 2 else None)
    first = (max(1, args.warmup // K) + 1) * world
    barrier()
    t0 = time.perf_counter()
    nbytes = 0
    for k in range(args.steps):
        nbytes += gop(enc, first + k * world + rank)
    barrier()
    dt = allmax(time.perf_counter() - t0)
    enc.close()
    ok = ok_oracle = None
    parts = [None] * world
    # ... and for the check against the CPU oracle's serial bytes (the first check compares the HIP path with itself): the
    # same sharding -- GOP g on rank g mod N, cedar_enc_set_frame_index -- at 320x240, where the scalar twin codes 2 N GOPs
    # in seconds. Every rank codes ITS OWN small GOPs on its own GPU; rank 0 only concatenates and compares.
    sw, sh = 320, 240
    small = {}
    with ce.CedarEncoder(width=sw, height=sh, qp=args.qp, keyframe_interval=K, device=local_rank) as sh_enc:
        for k in range(2):
            g = k * world + rank
            sh_enc.set_frame_index(g * K)
            small[g] = [sh_enc.encode_frame(*ce.gen_pattern("B", sw, sh, g * K + t)) for t in range(K)]
    small_parts = [None] * world
    if world > 1:
        dist.all_gather_object(parts, mine)
        dist.all_gather_object(small_parts, small)
    else:
        parts, small_parts = [mine], [small]
    if rank == 0:
        with ce.CedarEncoder(width=W, height=H, qp=args.qp, keyframe_interval=K, device=local_rank) as one:
            serial = []
            for f in range(2 * world * K):
                y, uv = inputs[f % nf]
                one.set_device_input(0, y.data_ptr(), uv.data_ptr())
                one.encode()
                serial.append(one.output(0))
        shards = []
        for k in range(2):
            for r in range(world):
                shards += parts[r][k * K:(k + 1) * K]
        ok = shards == serial
        if not ok:
            raise SystemExit("GOP shards from %d ranks do not concatenate to the serial stream" % world)
        from tests import oracle_lib as ol
        oe = ol.OracleEncoder(ol.default_config(sw, sh, qp=args.qp, keyint=K))
        oracle_serial = [oe.encode(*ol.gen_frame("B", sw, sh, f)) for f in range(2 * world * K)]
        got = []
        for g in range(2 * world):
            got += small_parts[g % world][g]
        ok_oracle = got == oracle_serial
        if not ok_oracle:
            raise SystemExit("GOP shards do not concatenate to the ORACLE's serial stream")
    if rank != 0:
        return None
    frames = args.steps * world * K
    value = frames * mbs_per_frame / dt
    return {"metric": "macroblocks/s @ fixed QP, 1080p nv12" if (W, H) == (1920, 1080) else "macroblocks/s @ fixed QP, %dx%d nv12" % (W, H),
            "value": value, "unit": "macroblocks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%dx%d nv12 I+P keyint %d QP %d, pattern-B, ONE stream sharded by closed GOPs: GOP g on GPU g mod %d "
                                   "(SURVEY 8(e) row 2; a step = one GOP of %d frames per GPU; no collective)" % (W, H, K, args.qp, world, K),
                       "parallelism": "closed GOPs x%d of one stream, no collective" % world,
                       "frames_per_step": world * K, "macroblocks_per_frame": mbs_per_frame,
                       "gop_shards_concatenate_to_the_serial_stream": ok,
                       "gop_shards_equal_the_cpu_oracles_serial_stream_320x240": ok_oracle,
                       "realtime_4k30_multiple": value / 972000.0},
            "roofline": None, "cpu_baseline": None}


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
