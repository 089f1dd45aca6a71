/*
 * include/cedar_h264enc.h -- C-ABI of the MI355X-native H.264 encode path.
 *
 * Drop-in boundary for the character-device session of libv/cedarx_h264_encoder
 * (reference: kernel/cedar_ioctl.h:1-48, kernel/cedar.c:1293-1300, caller
 * userspace/h264enc.c:47-117, 149, 189). The five verbs of that session
 *
 *     open("/dev/cedar_dev")  ioctl(CONFIG)  mmap x3  ioctl(ENCODE)  close
 *
 * map one-to-one to the entry points below. Plain C types only; the library
 * (libcedar_h264enc.so) owns every buffer, exactly as the driver does
 * (kernel/cedar.c:605-704), and hands back HOST pointers where the driver
 * hands back DMA addresses to mmap.
 *
 * What replaces the silicon (kernel/cedar.c:1176 STARTTRIG=0x08 .. :1178 IRQ
 * wait) is a per-macroblock HIP pipeline on gfx950; there is no CPU fallback:
 * every call fails with -ENODEV when no HIP device is usable.
 */
#ifndef CEDAR_H264ENC_H
#define CEDAR_H264ENC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* kernel/cedar_ioctl.h:15-16, 30-31 */
#define CEDAR_ENC_FORMAT_NV12 0
#define CEDAR_ENC_FORMAT_NV16 1 /* accepted by the reference's validation, never
				   programmed (cedar.c:1076); rejected here */
#define CEDAR_ENC_ENTROPY_CAVLC 0
#define CEDAR_ENC_ENTROPY_CABAC 1 /* reference default (h264enc.c:66); rejected here
				     with -EINVAL: north_star fixes baseline + CAVLC */

/*
 * Replaces struct cedar_ioctl_config (kernel/cedar_ioctl.h:12-46): same
 * fields, same order, same in/out meaning; the uint32_t *_dma_addr outputs
 * become host pointers valid until cedar_enc_close().
 */
struct cedar_enc_config {
	/* in */
	int src_width;          /* even (cedar.c:749)                        */
	int src_height;         /* even                                      */
	int src_format;         /* CEDAR_ENC_FORMAT_NV12                     */
	int dst_width;          /* multiple of 16, >= src (cedar.c:756-769)  */
	int dst_height;
	int profile;            /* profile_idc, 66                           */
	int level;              /* level_idc, 41 / 51                        */
	int qp;                 /* 1..47 (cedar.c:771)                       */
	int keyframe_interval;  /* 1..31 (cedar.c:784)                       */
	int thumbnail;          /* in: ignored; out: always 0 (out of scope) */
	int thumbnail_downscale;
	int entropy_coding_mode;/* CEDAR_ENC_ENTROPY_CAVLC                   */
	/* out (cedar.c:840-860) */
	uint8_t *input_luma;    /* src_width*src_height bytes per stream, tightly packed */
	int input_luma_size;    /* bytes per stream, ALIGN(.,4096) (cedar.c:612) */
	uint8_t *input_chroma;  /* src_width*src_height/2 bytes, interleaved CbCr */
	int input_chroma_size;
	uint8_t *bytestream;    /* Annex-B output of the last cedar_enc_encode() */
	int bytestream_size;    /* capacity per stream (reference: fixed 1 MiB, cedar.c:663) */
	uint8_t *thumb_luma;    /* always NULL / 0 */
	int thumb_luma_size;
	uint8_t *thumb_chroma;
	int thumb_chroma_size;
};

/*
 * Extensions with no counterpart in the reference (which is one device, one
 * session, one slice). Zero-initialise, set .size = sizeof, fill what you need.
 */
struct cedar_enc_config_ext {
	int size;            /* sizeof(struct cedar_enc_config_ext)                   */
	int device;          /* HIP device ordinal (default 0)                        */
	int n_streams;       /* lock-step streams of identical geometry batched into
				the same kernel launches (default 1). Stream s uses
				input_luma + s*input_luma_size, etc.                  */
	int slice_count;     /* slices per picture, bands of MB rows (default 1);
				> 1 sets disable_deblocking_filter_idc = 2            */
	int slice_first;     /* this session encodes slices [slice_first,             */
	int slice_num;       /*   slice_first+slice_num) only (0 = all): lets 4 GPUs
				each take one slice of a 4K picture, no collective    */
	int search_range;    /* full-pel motion search radius 1..16; 0 = default (16),
				-1 = zero-vector only                                 */
	int chroma_qp_offset_minus4; /* chroma_qp_index_offset - 4: 0 keeps the reference's
				+4 (cedar.c:971-972, :1165-1168)                      */
	int repeat_headers;  /* 1: SPS+PPS in front of every IDR, so closed GOPs encoded by
				different sessions / GPUs can be concatenated; 0: only
				before the very first frame (cedar.c:1058-1061)          */
	int aud;             /* 1: an access unit delimiter NAL (type 9) opens every access
				unit; 0: none, as the reference          */
	int fps_num;         /* fps_num, fps_den both > 0: the SPS carries VUI timing (fixed  */
	int fps_den;         /*   frame rate fps_num/fps_den) and a bitstream restriction
				(no reordering, one reference frame); 0/0: no VUI, as the
				reference (cedar.c:933)                                  */
	int first_frame_index; /* global index of the first frame this session encodes (default 0): a
				multiple of keyframe_interval. idr_pic_id and "SPS+PPS only in front of
				frame 0" derive from the GLOBAL index (cedar.c:1001, :1047, :1058-1061 count
				frames of the one device), so closed GOPs encoded by different sessions /
				GPUs concatenate to exactly the serial stream        */
	int gop_parallel;    /* G > 1: the session encodes G consecutive closed GOPs of every stream side by
				side (they depend on nothing of each other), as G lock-step lanes of the same
				kernel launches. Same bytes, same order as G = 1; the price is latency: frame
				n is delivered once frame n + (G-1)*keyframe_interval has been submitted (or
				after cedar_enc_flush). cedar_enc_session_depth() says how many frames the
				caller may have in flight. 0 / 1: off                                    */
	int strict_mb_bits;  /* H.264 A.3.1 allows a macroblock 3200 bits; pure noise at QP <= 16 passes that.
				0: such a picture is emitted as it is and counted
				   (cedar_enc_stats.oversize_frames);
				1: it fails with -EOVERFLOW and is dropped like any failed frame;
				2: it is coded again 6 QP steps coarser (slice_qp_delta), as often as it takes,
				   so no emitted macroblock ever exceeds the limit
				   (cedar_enc_stats.requantised_frames; one stream, whole pictures)       */
	int subpel;          /* motion vector precision: 0 whole samples (default; what round 1 shipped), 1 half,
				2 quarter samples -- a refinement of the 8 half- and then 8 quarter-sample
				neighbours of the best whole-sample vector, 6-tap interpolation of H.264
				8.4.2.2.1 (the VE keeps "sub-pel" planes per reference frame, cedar.c:520,
				cedar_regs.h:71-72; what the silicon does with them is not in the driver)  */
	int gop_low_latency; /* gop_parallel sessions. 0 (default): the caller may run 2 (G-1) keyframe_interval
				+ ring + 8 frames ahead, which puts the input of the NEXT group's IDR step in
				the session's hands while this group's last steps run: the IDR pictures hide
				beside them (8 lanes of 1080p: 84 -> 97 M macroblocks/s). 1: (G-1)
				keyframe_interval + ring only -- half the latency and buffering, every group's
				IDR step exposed. Same bytes either way                                    */
	int bitrate;         /* > 0: frame-level rate control towards this many bits per second (at fps_num /
				fps_den, 30 frames/s when unset). The VE has rate-control registers the
				reference never programs (cedar_regs.h:43-53); here every
				picture gets one QP, carried by slice_qp_delta, from a law that only looks at
				frames at least CEDAR_ENC_PIPELINE_DEPTH + 1 back -- so the bytes do not
				depend on how many frames the caller keeps in flight. qp is the starting
				point. With n_streams > 1 the law runs per stream (every stream towards this
				target, its own QP per picture). With gop_parallel = G (one stream) the law
				is per group of G GOPs -- one QP for the group, from the mean bits per frame
				of the group 1 + ceil(16 / keyframe_interval) groups back, the nearest one
				every step of which has left the session by then. All slices.
				0: fixed QP, as the reference (cedar.c:1165-1168)                             */
	int intra_refresh;   /* N >= 3: cyclic intra refresh -- the VE has a register for it that the reference
				never writes (cedar_regs.h:43 CEDAR_H264ENC_CYCLICINTRAREFRESH, SURVEY
				8(f)-4). In the t-th P picture after an IDR the macroblocks with
				(x + 2*y + t) % N == 0 are coded Intra16x16 (DC), so any N consecutive P
				pictures renew every macroblock; the PPS then carries
				constrained_intra_pred_flag = 1 (a refresh macroblock must not lean on inter
				samples, or it would refresh nothing). 0 (default): off, PPS as the
				reference's (cedar.c:976)                                                  */
	int roi_count;       /* 0..4 regions of interest: the VE has four plus their QP offsets
				(cedar_regs.h:76-80 ROI_QP_OFFSET, ROI_0..3_AREA) that the reference never
				programs                                                   */
	int roi[4][5];       /* x, y, width, height in macroblocks and the QP offset (-24..24) of the
				macroblocks inside; where regions overlap the first counts. A macroblock's QP
				= the picture's (qp, or what rate control / requantising make of it) + offset,
				clipped to 1..51, carried by mb_qp_delta. In such a session the first
				macroblock of every row always carries one (an empty chroma DC pair if it has
				nothing else to code): QP_Y is then never inherited across a row boundary     */
	int mad_th[4];       /* MAD classes -- the VE has four MAD thresholds and a MAD sum for its rate control
				(cedar_regs.h:46-50 RC_MAD_TH0..3, RC_MAD_SUM, MAD), none of it programmed by
				the reference. Four ascending thresholds 0..255 on the macroblock's mean absolute
				difference against the prediction it chose (luma SAD / 256: the motion-compensated
				block, the chosen intra kind in IDR pictures) make five classes ...            */
	int mad_dqp[5];      /* ... and each class moves the macroblock's QP by this much (-12..12) on top of the
				picture's QP and its region's: finer where prediction works (what the eye follows),
				coarser where it fails anyway, or the other way round. Decisions (lambda) are taken
				at the QP without it. All zero (default): off. The row rule of roi applies          */
	int pcm;             /* 0 (default): a macroblock whose macroblock_layer() could pass the 3200 bits H.264
				A.3.1 allows is coded I_PCM -- its source samples, raw, 3088 bits -- so that no
				emitted macroblock exceeds the limit (the VE's silicon behind cedar.c:1176 has to
				honour it at every QP cedar.c:771-775 accepts). The sessions and pictures that
				take it: one QP per picture (no roi, no MAD classes, one stream's rate control),
				no intra_refresh, and a picture QP at which the edges of an I_PCM macroblock --
				QP_Y 0 for the deblocking filter -- are not filtered at all: up to 22 with the
				reference's chroma offset (noise passes 3200 bits up to QP 16). Elsewhere, and with
				-1 (never), strict_mb_bits says what happens                                       */
	int scene_cut;       /* 1..100: a P picture in which at least this percentage of the macroblocks is predicted
				better by its own mean than by the best vector (SAD against the mean + 32 lambda
				< the search's cost) is coded as an IDR picture instead and starts a new GOP
				there -- what a scene cut wants is the I-picture path, all modes, for that
				picture (the VE's MEPARA / PARA0, cedar.c:1155-1171, leave intra-in-P to
				silicon). The analysis reports the count with the frame; the session codes the
				frame again as an IDR picture, and the frames in flight behind it after it (the
				machinery of a failed frame). One stream, whole pictures, no gop_parallel. 0: off   */
};

typedef struct cedar_enc cedar_enc;

/* open(CEDAR_DEVICE_PATH, O_RDWR) -- h264enc.c:149, cedar.c:457-474.
 * Returns NULL when no usable HIP device exists (errno = ENODEV). */
cedar_enc *cedar_enc_open(void);

/* ioctl(fd, CEDAR_IOCTL_CONFIG, &cfg) -- cedar.c:732-866. Returns 0 or a
 * negative errno: -EINVAL (bad parameter, NULL, already configured), -ENOMEM,
 * -ENODEV. One-shot per open (cedar.c:744-747). */
int cedar_enc_config(cedar_enc *enc, struct cedar_enc_config *cfg);
int cedar_enc_config_ext(cedar_enc *enc, struct cedar_enc_config *cfg,
			 const struct cedar_enc_config_ext *ext);

/* ioctl(fd, CEDAR_IOCTL_ENCODE, NULL) -- cedar.c:1032-1209. Encodes the frame
 * currently in the input buffers (of every stream). Synchronous. Returns the
 * number of bytes now valid at bytestream offset 0 of stream 0 (n_streams == 1:
 * the reference contract), or the total over all streams when batching; < 0 is a
 * negative errno and leaves GOP counters and reference frames untouched
 * (cedar.c:1182-1191 return before :1193-1201). Output of call n =
 * [SPS PPS if n == 0] + one slice NAL per encoded slice (cedar.c:1058-1066). */
int cedar_enc_encode(cedar_enc *enc);

/* close(fd) -- cedar.c:706-730 */
void cedar_enc_close(cedar_enc *enc);

/* ---- extensions (no reference counterpart) ---- */

/* per-stream result of the last encode: pointer into the bytestream buffer */
int cedar_enc_stream_output(cedar_enc *enc, int stream, const uint8_t **data, int *size);

/* Use frames already resident in HBM as the next input of `stream` (tightly
 * packed src_width x src_height nv12, device pointers on the session's device)
 * instead of copying from the pinned input buffers. NULL restores the default. */
int cedar_enc_set_device_input(cedar_enc *enc, int stream, const void *d_luma, const void *d_chroma);

/* Pipelined form of encode for throughput: submit queues one frame for every
 * stream without waiting; collect waits for the oldest submitted frame and
 * returns its byte count (same meaning as cedar_enc_encode). cedar_enc_encode ==
 * submit + collect. A session allows cedar_enc_session_depth() frames in flight
 * (submit returns -EBUSY beyond that): 4, or CEDAR_ENC_PIPELINE_DEPTH (the
 * maximum) for sessions of few streams, whose frames overlap on the device --
 * frame n+1 is analysed behind frame n's deblocking wavefront, and an IDR
 * picture, which depends on no other frame, runs beside the P pictures submitted
 * before it, so it costs no time when the caller keeps the ring full. The bytes
 * are the same at any depth. Device input frames (cedar_enc_set_device_input)
 * must stay untouched until their frame is collected.
 *
 * Failure (the reference: cedar.c:1182-1201 returns before the GOP counter and the reference
 * swap, and h264enc.c:189-192 goes on with the next frame): a collect that returns < 0 has
 * dropped exactly that frame. GOP counters and reference frames are as they were before it, and
 * the frames submitted after it are encoded again by the session itself, from the inputs it
 * still holds (its own device copies; device input frames, which the caller keeps untouched
 * until their frame is collected anyway) -- the next collect delivers the frame after the failed
 * one, coded as if the failed one had never been submitted.
 *
 * Outputs (cedar_enc_stream_output, cfg->bytestream) stay valid until the next cedar_enc_submit
 * or cedar_enc_encode on the session: the ring slot they live in is then written again. */
#define CEDAR_ENC_PIPELINE_DEPTH 16
int cedar_enc_pipeline_depth(void);                  /* the compile-time maximum (gop_parallel = 0) */
int cedar_enc_session_depth(const cedar_enc *enc);   /* this session's; -EINVAL before CONFIG */
int cedar_enc_submit(cedar_enc *enc);
int cedar_enc_collect(cedar_enc *enc);
/* The next frame submitted is frame `index` of the stream (a multiple of keyframe_interval) and starts
 * a GOP: a session that codes GOPs r, r + N, r + 2N, ... of a stream sharded over N GPUs moves on with
 * this instead of being configured again (first_frame_index is where it starts). idr_pic_id and the
 * SPS / PPS cadence follow the index (cedar.c:1001, :1047, :1058-1061). Not for gop_parallel sessions. */
int cedar_enc_set_frame_index(cedar_enc *enc, int index);
/* gop_parallel sessions: no more frames will come; the GOPs buffered so far are encoded with the
 * missing lanes idle, and cedar_enc_collect delivers what is left. collect returns -EAGAIN when
 * the next frame in order needs more input (or a flush) first. A no-op for other sessions. */
int cedar_enc_flush(cedar_enc *enc);

/* Timing of the kernels, measured with hipEvents on the stream they run on
 * (replaces the driver's "Time spent" accounting, cedar.c:715-719, :1203-1205). */
enum cedar_enc_stage {
	CEDAR_ENC_STAGE_INPUT = 0,   /* H2D + pad                      */
	CEDAR_ENC_STAGE_INTRA_DECIDE,
	CEDAR_ENC_STAGE_INTRA_RECON,
	CEDAR_ENC_STAGE_INTER,       /* ME + residual + recon          */
	CEDAR_ENC_STAGE_DEBLOCK,
	CEDAR_ENC_STAGE_ENTROPY,     /* CAVLC + scan + pack + escape   */
	CEDAR_ENC_STAGE_COUNT
};
struct cedar_enc_stats {
	uint64_t frames;                         /* frames encoded (per stream)   */
	uint64_t launches[CEDAR_ENC_STAGE_COUNT];/* kernel launches per stage     */
	double   ms[CEDAR_ENC_STAGE_COUNT];      /* summed GPU time per stage     */
	double   encode_wall_ms;                 /* host wall time inside encode  */
	int      wf_split;                       /* workgroups (CUs) per unit in the wavefront kernels */
	int      deblock_planes_split;           /* 1: luma and chroma deblock run as two kernels      */
	int      frames_pipelined;               /* 1: frames in flight overlap on the device          */
	int      idr_lookahead;                  /* 1: IDR pictures run on streams of their own        */
	int      depth;                          /* frames in flight this session allows               */
	int      deblock_streams;                /* deblock streams in use                             */
	uint64_t oversize_frames;                /* frames with a macroblock above 3200 bits (A.3.1)   */
	int      frame_qp;                       /* slice QP of the frame collected last (rate control; of stream 0 when several streams are steered) */
	uint64_t requantised_frames;             /* re-encodes because of a macroblock above 3200 bits  */
	uint64_t scene_cuts;                     /* P pictures coded again as IDR pictures (scene_cut)  */
	int      deblock_rows;                   /* 1: deblocking with a wave per macroblock row, a workgroup per band of 4 rows */
	int      intra_rows;                     /* 1: intra reconstruction as a wavefront of 4x4 blocks, a wave per macroblock row */
	int      src_direct;                     /* 1: the last frame was read where the caller put it (no padded copy, no k_input) */
	int      entropy_fused;                  /* 1: k_pack_out behind k_scan (3 launches), 0: the round-3 chain of 8 */
};
/* event timing: 0 off (default), 1 every stage, else a mask with bit (1 + stage) per stage to time;
 * bits 8..15 = n: time every n-th frame only (0/1: every frame) */
int cedar_enc_set_profiling(cedar_enc *enc, int on);
int cedar_enc_get_stats(cedar_enc *enc, struct cedar_enc_stats *out);
int cedar_enc_reset_stats(cedar_enc *enc);

/* test hook: copy an internal device array of the last encoded frame to host.
 * Returns bytes copied or negative errno. */
enum cedar_enc_debug_buf {
	CEDAR_ENC_DBG_MBINFO = 0, /* struct { int8 type,i16,chroma,cbp; int16 mvx,mvy (quarter samples); } per MB */
	CEDAR_ENC_DBG_I4MODES,    /* int8[16] per MB, coding-order index                       */
	CEDAR_ENC_DBG_NNZ,        /* uint8[24] per MB                                          */
	CEDAR_ENC_DBG_COEF,       /* int16[408] per MB: luma[16][16] dc16[16] cdc[2][4] cac[2][4][16] */
	CEDAR_ENC_DBG_RECON_Y,    /* deblocked reconstruction, coded size                      */
	CEDAR_ENC_DBG_RECON_UV,
	CEDAR_ENC_DBG_SRC_Y,      /* padded source                                             */
	CEDAR_ENC_DBG_SRC_UV,
	CEDAR_ENC_DBG_MB_BITS,    /* uint32 bits per MB                                        */
	CEDAR_ENC_DBG_MB_QP       /* uint8 per MB: QP_Y the decoder holds for it (sessions with roi / mad_dqp; -ENOENT otherwise) */
};
int cedar_enc_debug_read(cedar_enc *enc, int what, int stream, void *dst, int dst_size);

/* synthetic nv12 test patterns 'A','B','C' (SURVEY.md 8(d)); host helper for the
 * CLI/bench, independent of the oracle's generator */
void cedar_enc_gen_pattern(int pattern, int width, int height, int frame,
			   unsigned seed_offset, uint8_t *luma, uint8_t *chroma);

const char *cedar_enc_last_error(cedar_enc *enc);
const char *cedar_enc_version(void);

/* Schedule introspection (no device needed; tests/test_plan.py): what CONFIG decides for a geometry -- frames in
 * flight, how the two wavefront kernels are split over workgroups, the deblock form, whether frames are pipelined on
 * the device, stripes and how many are held back, cohorts -- given what the stream probes found. The CEDAR_*
 * development switches are read from the environment as CONFIG reads them. Returns 0, or -EINVAL. */
struct cedar_enc_plan_query {
	int mb_w, mb_h;                         /* coded size in macroblocks */
	int slice_count, slice_first, slice_num; /* as cedar_enc_config_ext (slice_num 0 = all) */
	int n_streams;                          /* lock-step streams x GOP lanes */
	int cus;                                /* compute units of the device (MI355X: 256) */
	int db_streams_usable;                  /* probe result: deblock streams that run beside the others */
	int holds_pipe_token;                   /* 1: no other pipelined session on the device in this process */
};
struct cedar_enc_plan_result {
	int error;                              /* 1: a slice too tall for the deblock wavefront */
	int depth, wf_split, wf_split_intra, db_pairs, split_planes;
	int want_row_pipe, db_streams_wanted;   /* before the probes */
	int row_pipe, db_streams, inter_bands, defer_default, cohorts; /* after them */
	int deblock_rows;                       /* 1: deblocking with a wave per macroblock row (few units), 0: wave pairs per band */
	int intra_rows;                         /* 1: intra reconstruction as a wavefront of 4x4 blocks, a wave per macroblock row */
	/* who may be resident and waiting (sessions that pipeline on k_deblock_rows): the workgroups a picture's deblocking
	 * puts on ONE of the 8 XCDs (each XCD places its share of a launch in order on its own CUs), the KiB of LDS each
	 * claims, and whether a third picture's -- an IDR picture's deblocking beside its intra wavefront -- may join the two */
	int db_wgs_per_xcd, db_lds_kb, idr_beside_ok;
};
int cedar_enc_plan(const struct cedar_enc_plan_query *q, struct cedar_enc_plan_result *r);

#ifdef __cplusplus
}
#endif

#endif /* CEDAR_H264ENC_H */
