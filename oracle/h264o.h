/*
 * oracle/h264o.h -- scalar C restatement of the encode hot path (the oracle).
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg as the checker. The product path
 * (cedarx_h264_encoder_amd/) never includes, links or calls anything here.
 *
 * What it restates, with the reference lines it follows:
 *   - session lifecycle / validation      kernel/cedar.c:732-866
 *   - GOP cadence, frame_num, ref swap    kernel/cedar.c:1047-1050, 1193-1201
 *   - put-bits, ue(v), se(v)              kernel/cedar.c:187-223
 *   - NAL start code, SPS, PPS, slice hdr kernel/cedar.c:868-1030
 *   - frame pump                          userspace/h264enc.c:119-201
 * The macroblock loop itself is fixed-function silicon in the reference
 * (kernel/cedar.c:1176 STARTTRIG=0x08 .. :1178 wait for IRQ): it has no
 * software form, so that part follows ITU-T H.264 (clauses 7.3, 8.3, 8.4,
 * 8.5, 8.7, 9.1, 9.2) with the encoder-side policy documented in DESIGN.md.
 *
 * PARITY UNPINNED at macroblock level (the reference holds no tests, golden
 * vectors or runnable software for it). Header bytes ARE pinned by the
 * known-answer vectors of SURVEY.md 8(c) (tests/golden/headers.json).
 */
#ifndef H264O_H
#define H264O_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { H264O_MB_I4x4 = 0, H264O_MB_I16x16 = 1, H264O_MB_P16x16 = 2, H264O_MB_PSKIP = 3,
       H264O_MB_IPCM = -1 /* intra like 0 and 1 ("<= I16x16"), no refIdx like them (">= P16x16" is false) */ };

typedef struct h264o_config {
	int width;            /* source width, even (cedar.c:749)  */
	int height;           /* source height, even               */
	int qp;               /* 1..47 (cedar.c:771)               */
	int keyint;           /* 1..31 (cedar.c:784)               */
	int profile_idc;      /* 66 baseline                       */
	int level_idc;        /* 41; 51 for 4K                     */
	int search_range;     /* full-pel, 0..16                   */
	int slice_count;      /* >= 1, bands of MB rows            */
	int chroma_qp_offset; /* 4 (cedar.c:971, :1167)            */
	int repeat_headers;   /* 1: SPS+PPS in front of every IDR (reference: frame 0 only, cedar.c:1058) */
	int aud;              /* 1: access unit delimiter NAL (type 9) opens every access unit */
	int fps_num, fps_den; /* both > 0: the SPS carries VUI timing (fixed frame rate fps_num/fps_den)
	                       * and a bitstream restriction (no reordering, 1 reference frame);
	                       * the reference writes vui_parameters_present_flag = 0 (cedar.c:933) */
	int subpel;           /* motion vector precision: 0 whole samples (default), 1 half, 2 quarter -- a
	                       * refinement around the best whole-sample vector (the VE keeps sub-pel planes,
	                       * cedar.c:520, cedar_regs.h:71-72; what it does with them is not in the driver) */
	int bitrate;          /* > 0: frame-level rate control towards this many bits per second (frame rate
	                       * fps_num/fps_den, 30 when unset): the slice QP follows h264o_rc_pick_qp;
	                       * 0: fixed QP, as the reference (cedar.c:1165-1168 PARA1 = qp) */
	int strict_mb_bits;   /* a picture with a macroblock_layer() of more than 3200 bits (H.264 A.3.1): 0 emitted
	                       * and counted; 1 refused with -EOVERFLOW; 2 coded again 6 QP steps coarser (slice_qp_delta)
	                       * until every macroblock fits, -EOVERFLOW if QP 51 does not do it */
	int intra_refresh;    /* N >= 3: cyclic intra refresh (the VE has a register for it, cedar_regs.h:43
	                       * CYCLICINTRAREFRESH, that the driver never writes) -- in the t-th P picture after an IDR
	                       * the macroblocks with (x + 2*y + t) % N == 0 are coded Intra16x16 (DC), and the PPS sets
	                       * constrained_intra_pred_flag so that they do not lean on inter samples; 0: off */
	int roi_count;        /* 0..4 regions of interest (the VE has four, cedar_regs.h:76-80 ROI_QP_OFFSET, ROI_n_AREA;
	                       * the driver never writes them) */
	int roi[4][5];        /* x, y, width, height in macroblocks and the QP offset (-24..24) of the macroblocks inside;
	                       * where regions overlap the first one counts. QP of a macroblock = slice QP + offset,
	                       * clipped to 1..51, carried by mb_qp_delta. With regions (or MAD classes) the first macroblock
	                       * of every row always carries an mb_qp_delta */
	int mad_th[4];        /* MAD classes (the VE has four thresholds, cedar_regs.h:46-49 RC_MAD_TH0..3, and a MAD sum; the
	                       * driver programs none of it): ascending thresholds 0..255 on the macroblock's mean absolute
	                       * difference against its chosen prediction (luma SAD / 256) ... */
	int mad_dqp[5];       /* ... and the QP offset (-12..12) of each of the five classes, on top of the picture's QP and
	                       * the region's. All zero: off */
	int scene_cut;        /* 1..100: a P picture in which at least this percentage of the macroblocks is better predicted by
	                       * its own mean than by the best vector (SAD against the mean + 32 lambda < the search's cost) is
	                       * coded as an IDR picture instead and starts a new GOP there; 0: off. One stream, whole pictures */
	int pcm;              /* 0 (default): a macroblock whose macroblock_layer() could pass the 3200 bits of H.264 A.3.1 is
	                       * coded I_PCM -- its source samples, raw -- in the sessions that take it (one QP per picture of
	                       * at most 30, no cyclic refresh: DESIGN.md 3.6); -1: never, strict_mb_bits says what happens */
	int rc_lanes;         /* > 1 (with bitrate): the rate control law of a session that codes this many closed GOPs side by side
	                       * (gop_parallel of the HIP path): one QP per GROUP of rc_lanes GOPs, taken from the bits of the
	                       * group 1 + ceil(16 / keyint) groups back -- the nearest one every frame of which has left a
	                       * session with 16 steps in flight; qp before that. 0, 1: the law by frames (bitrate above) */
} h264o_config;

/* per-macroblock record kept for parity debugging against the HIP path */
typedef struct h264o_mb {
	int8_t  type;         /* H264O_MB_*                                      */
	int8_t  i16_mode;     /* 0 V 1 H 2 DC 3 Plane                            */
	int8_t  chroma_mode;  /* 0 DC 1 H 2 V 3 Plane                            */
	int8_t  cbp;          /* bits 0-3 luma 8x8, bits 4-5 chroma (0,1,2)      */
	int8_t  i4_modes[16]; /* by coding-order block index                     */
	int16_t mvx, mvy;     /* quarter samples                                 */
	uint8_t nnz[24];      /* total_coeff: 16 luma (raster bx+4*by), 4 Cb, 4 Cr */
	int16_t luma[16][16]; /* levels, zig-zag order, coding-order block index */
	int16_t dc16[16];     /* Intra16x16 DC levels, zig-zag                   */
	int16_t cdc[2][4];    /* chroma DC levels                                */
	int16_t cac[2][4][16];/* chroma AC levels, zig-zag, [.][.][0] unused     */
} h264o_mb;

typedef struct h264o_enc h264o_enc;

void h264o_default_config(h264o_config *cfg, int width, int height);
/* returns NULL and sets *err (negative errno) on invalid parameters */
h264o_enc *h264o_open(const h264o_config *cfg, int *err);
void h264o_close(h264o_enc *e);

/* Encode one frame. luma: width*height bytes, chroma: width*height/2 bytes of
 * interleaved CbCr, both tightly packed (userspace/h264enc.c:178-187).
 * Returns the number of bytes written to out, or a negative errno. */
int h264o_encode(h264o_enc *e, const uint8_t *luma, const uint8_t *chroma,
		 uint8_t *out, int out_cap);

/* worst-case output bytes per frame for out_cap */
int h264o_max_frame_bytes(const h264o_enc *e);

/* state of the last encoded frame */
int h264o_mb_width(const h264o_enc *e);
int h264o_mb_height(const h264o_enc *e);
const h264o_mb *h264o_mbs(const h264o_enc *e);
/* reconstructed (deblocked) picture of the last frame, coded size, nv12 */
const uint8_t *h264o_recon_luma(const h264o_enc *e);
const uint8_t *h264o_recon_chroma(const h264o_enc *e);
int h264o_last_frame_type(const h264o_enc *e); /* 1 = IDR, 0 = P */
int h264o_frame_qp(const h264o_enc *e);        /* slice QP of the last frame (rate control) */
int h264o_max_mb_bits(const h264o_enc *e);     /* largest macroblock_layer() of the last frame, in bits */
const uint8_t *h264o_mb_qp(const h264o_enc *e); /* [mb] QP_Y the decoder holds for every macroblock of the last frame */

/* stand-alone pieces exported for known-answer tests */
int h264o_write_sps(const h264o_config *cfg, uint8_t *out, int cap);
int h264o_write_pps(const h264o_config *cfg, uint8_t *out, int cap);
int h264o_write_aud(int idr, uint8_t *out, int cap);
/* slice header only (start code, NAL byte, header bits, zero-padded to a byte);
 * *nbits receives the number of RBSP bits after the NAL header byte */
int h264o_write_slice_header(const h264o_config *cfg, int idr, int frame_num,
			     int idr_pic_id, int first_mb, uint8_t *out,
			     int cap, int *nbits);
/* CAVLC of one residual block: levels in zig-zag order, max_coeff 16/15/4,
 * nC context (-1 = chroma DC). Returns number of bits written MSB-first. */
int h264o_cavlc_block(const int16_t *levels, int max_coeff, int nC,
		      uint8_t *out, int cap);
void h264o_fdct4x4(const int16_t in[16], int16_t out[16]);
void h264o_idct4x4_add(const int16_t coef[16], uint8_t *dst, int stride);
int h264o_ue_len(unsigned v);

/* synthetic inputs of SURVEY.md 8(d); pattern 'A', 'B' or 'C' */
void h264o_gen_frame(int pattern, int width, int height, int frame,
		     unsigned seed_offset, uint8_t *luma, uint8_t *chroma);

/* ---- test-only conformance decoder (oracle/h264o_dec.c) ---- */
typedef struct h264o_dec h264o_dec;
h264o_dec *h264o_dec_open(void);
void h264o_dec_close(h264o_dec *d);
/* Feed one access unit (any number of NALs). Returns number of pictures
 * completed (0 or 1) or a negative value on a syntax error. */
int h264o_dec_decode(h264o_dec *d, const uint8_t *data, int size);
int h264o_dec_width(const h264o_dec *d);   /* coded width  (multiple of 16) */
int h264o_dec_height(const h264o_dec *d);
int h264o_dec_crop_width(const h264o_dec *d);  /* after SPS cropping */
int h264o_dec_crop_height(const h264o_dec *d);
const uint8_t *h264o_dec_luma(const h264o_dec *d);   /* coded size, stride = width */
const uint8_t *h264o_dec_chroma(const h264o_dec *d); /* nv12 interleaved */
const char *h264o_dec_error(const h264o_dec *d);
void h264o_dec_frame_rate(const h264o_dec *d, unsigned *num, unsigned *den);
int h264o_dec_reorder_frames(const h264o_dec *d); /* -1: no bitstream restriction in the SPS */
int h264o_dec_aud_count(const h264o_dec *d);

#ifdef __cplusplus
}
#endif
#endif /* H264O_H */
