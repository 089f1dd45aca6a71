/*
 * oracle/h264o_enc.c -- scalar C H.264 baseline encoder: the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY (see h264o.h). PARITY UNPINNED at macroblock level.
 *
 * Structure (mirrors what the HIP path must reproduce bit for bit):
 *   1. pad source to macroblock multiples
 *   2. analysis + reconstruction per macroblock
 *        I pictures: open-loop mode decision on SOURCE samples (parallel on a
 *                    GPU), then closed-loop prediction/transform/quant/recon
 *        P pictures: full-pel 16x16 SAD search against the deblocked previous
 *                    reconstruction, P_L0_16x16 or P_Skip, no intra in P
 *   3. in-loop deblocking (macroblock raster order, clause 8.7)
 *   4. CAVLC entropy coding per slice, emulation prevention, NAL framing
 *
 * Reference lines followed for the session shell: kernel/cedar.c:732-866
 * (config validation), :868-1030 (headers), :1047-1050 and :1193-1201 (GOP).
 */
#include <errno.h>
#include <stdlib.h>
#include <string.h>

#include "h264o.h"
#include "h264o_tables.h"

#define ALIGN16(x) (((x) + 15) & ~15)
#define SEARCH_GRID 33 /* candidate index grid is fixed 33x33 (+-16) */

struct h264o_enc {
	h264o_config cfg;
	int mb_w, mb_h, cw, ch;
	int frame_count;   /* cedar.c:95  */
	int frame_p_count; /* cedar.c:94  */
	uint8_t *src_y, *src_uv;
	uint8_t *rec_y[2], *rec_uv[2];
	int cur; /* index of reference_current (cedar.c:117-119) */
	h264o_mb *mbs;
	int last_idr;
	int idr_count;   /* IDR pictures coded so far (cfg.scene_cut: idr_pic_id alternates with it) */
	int cut_votes;   /* cfg.scene_cut: macroblocks of the P picture being analysed that their own mean predicts better */
	int qp;          /* slice QP of the frame being coded: cfg.qp, or what the rate control picks */
	int mbqp;        /* QP of the macroblock being analysed: qp + the offset of its region of interest (+ its MAD class) */
	int mad_on;      /* cfg.mad_dqp has a non-zero entry */
	uint8_t *qp_t;   /* [mb] QP every macroblock is quantised with (qp when no region covers it) */
	uint8_t *qp_eff; /* [mb] QP_Y the decoder holds for it (7.4.5: a macroblock without mb_qp_delta keeps its predecessor's) */
	int64_t rc_bits[64]; /* rate control history, index = frame number % 64: bits of the access unit ... */
	int rc_qp[64];       /* ... and the QP it was coded at */
	int64_t grp_sum[32]; /* rc_lanes: per group of rc_lanes GOPs (index = group % 32) the bits of its frames, each scaled to the configured QP ... */
	int grp_n[32];       /* ... how many frames that is ... */
	long grp_id1[32];    /* ... and which group it is, + 1 (0: none yet) */
	int closed_loop; /* env H264O_CLOSED_LOOP_INTRA=1 (tools/rd_record.py only): intra modes decided on reconstructed neighbours */
	int max_mb_bits; /* largest macroblock_layer() of the last frame (A.3.1: at most 3200 bits) */
	uint8_t cbp_code_intra[48], cbp_code_inter[48];
	uint8_t *rbsp; /* scratch for one slice's RBSP */
	int rbsp_cap;
};

/* ------------------------------------------------------------------ */
/* bit writer: cedar_bytestream_write (cedar.c:187-207) without the HW */

typedef struct bitw {
	uint8_t *buf;
	int cap;   /* bytes */
	long pos;  /* bits  */
	int overflow;
} bitw;

static void bw_init(bitw *b, uint8_t *buf, int cap)
{
	b->buf = buf;
	b->cap = cap;
	b->pos = 0;
	b->overflow = 0;
	if (cap > 0)
		memset(buf, 0, (size_t)cap);
}

/* append the low n bits of v, MSB first, n <= 32 */
static void bw_put(bitw *b, uint32_t v, int n)
{
	int i;

	for (i = n - 1; i >= 0; i--) {
		long byte = b->pos >> 3;

		if (byte >= b->cap) {
			b->overflow = 1;
			return;
		}
		if ((v >> i) & 1)
			b->buf[byte] |= (uint8_t)(0x80 >> (b->pos & 7));
		b->pos++;
	}
}

static int bitlen32(uint32_t v)
{
	int n = 0;

	while (v) {
		n++;
		v >>= 1;
	}
	return n;
}

int h264o_ue_len(unsigned v)
{
	return 2 * bitlen32(v + 1) - 1;
}

/* cedar_bytestream_expgolomb (cedar.c:209-215) */
static void bw_ue(bitw *b, uint32_t v)
{
	v++;
	bw_put(b, v, 2 * bitlen32(v) - 1);
}

/* cedar_bytestream_expgolomb_signed (cedar.c:217-223): k>0 -> 2k-1, k<=0 -> -2k */
static void bw_se(bitw *b, int32_t v)
{
	bw_ue(b, v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v));
}

/* rbsp_trailing_bits, standard form: stop bit then zeros to a byte boundary.
 * (cedar.c:883-890 emits 9 bits when already one short of alignment: a stray
 * 0x00 byte, which a NAL must not end with; documented divergence.) */
static void bw_trailing(bitw *b)
{
	bw_put(b, 1, 1);
	while (b->pos & 7)
		bw_put(b, 0, 1);
}

/* NAL framing: 4-byte start code (cedar.c:875-878) + header byte + payload with
 * emulation prevention (done by the silicon in the reference, cedar.c:872-880) */
static int nal_write(uint8_t *out, int cap, int ref_idc, int type,
		     const uint8_t *rbsp, int n)
{
	int o = 0, zeros = 0, i;

	if (cap < 5)
		return -ENOSPC;
	out[o++] = 0;
	out[o++] = 0;
	out[o++] = 0;
	out[o++] = 1;
	out[o++] = (uint8_t)(((ref_idc & 3) << 5) | (type & 0x1f));
	for (i = 0; i < n; i++) {
		if (zeros == 2 && rbsp[i] <= 3) {
			if (o >= cap)
				return -ENOSPC;
			out[o++] = 3;
			zeros = 0;
		}
		if (o >= cap)
			return -ENOSPC;
		out[o++] = rbsp[i];
		zeros = rbsp[i] == 0 ? zeros + 1 : 0;
	}
	return o;
}

/* ------------------------------------------------------------------ */
/* parameter sets and slice header (cedar.c:892-1030)                  */

static void sps_rbsp(const h264o_config *c, bitw *b)
{
	int cw = ALIGN16(c->width), ch = ALIGN16(c->height);

	bw_put(b, (uint32_t)c->profile_idc, 8);
	bw_put(b, 0, 8); /* constraint flags: whole byte 0 (cedar.c:898-899) */
	bw_put(b, (uint32_t)c->level_idc, 8);
	bw_ue(b, 0); /* seq_parameter_set_id */
	bw_ue(b, 0); /* log2_max_frame_num_minus4 */
	bw_ue(b, 2); /* pic_order_cnt_type */
	bw_ue(b, 1); /* max_num_ref_frames */
	bw_put(b, 0, 1); /* gaps_in_frame_num_value_allowed_flag */
	bw_ue(b, (uint32_t)(cw / 16 - 1));
	bw_ue(b, (uint32_t)(ch / 16 - 1));
	bw_put(b, 1, 1); /* frame_mbs_only_flag */
	bw_put(b, 0, 1); /* direct_8x8_inference_flag */
	if (cw != c->width || ch != c->height) {
		/* crop offsets in units of 2 luma samples for 4:2:0 frames
		 * (cedar.c:924-931 writes raw pixel counts: a latent bug) */
		bw_put(b, 1, 1);
		bw_ue(b, 0);
		bw_ue(b, (uint32_t)((cw - c->width) / 2));
		bw_ue(b, 0);
		bw_ue(b, (uint32_t)((ch - c->height) / 2));
	} else {
		bw_put(b, 0, 1);
	}
	if (c->fps_num > 0 && c->fps_den > 0) {
		/* H.264 E.1.1: timing + bitstream restriction, nothing else */
		bw_put(b, 1, 1); /* vui_parameters_present_flag */
		bw_put(b, 0, 4); /* aspect_ratio, overscan, video_signal_type, chroma_loc: absent */
		bw_put(b, 1, 1); /* timing_info_present_flag */
		bw_put(b, (uint32_t)c->fps_den, 32);     /* num_units_in_tick */
		bw_put(b, (uint32_t)c->fps_num * 2u, 32); /* time_scale: two ticks (fields) per frame */
		bw_put(b, 1, 1); /* fixed_frame_rate_flag */
		bw_put(b, 0, 3); /* nal_hrd, vcl_hrd, pic_struct: absent */
		bw_put(b, 1, 1); /* bitstream_restriction_flag */
		bw_put(b, 1, 1); /* motion_vectors_over_pic_boundaries_flag */
		bw_ue(b, 0);     /* max_bytes_per_pic_denom: no limit */
		bw_ue(b, 0);     /* max_bits_per_mb_denom: no limit */
		bw_ue(b, 7);     /* log2_max_mv_length_horizontal: |mv| <= 16 pel = 64 quarter-pel */
		bw_ue(b, 7);     /* log2_max_mv_length_vertical */
		bw_ue(b, 0);     /* max_num_reorder_frames: output order = decode order */
		bw_ue(b, 1);     /* max_dec_frame_buffering */
	} else {
		bw_put(b, 0, 1); /* vui_parameters_present_flag (cedar.c:933) */
	}
	bw_trailing(b);
}

static void pps_rbsp(const h264o_config *c, bitw *b)
{
	bw_ue(b, 0); /* pic_parameter_set_id */
	bw_ue(b, 0); /* seq_parameter_set_id */
	bw_put(b, 0, 1); /* entropy_coding_mode_flag: CAVLC (cedar.c:949-952) */
	bw_put(b, 0, 1); /* bottom_field_pic_order_in_frame_present_flag */
	bw_ue(b, 0); /* num_slice_groups_minus1 */
	bw_ue(b, 0); /* num_ref_idx_l0_default_active_minus1 */
	bw_ue(b, 0); /* num_ref_idx_l1_default_active_minus1 */
	bw_put(b, 0, 1); /* weighted_pred_flag */
	bw_put(b, 0, 2); /* weighted_bipred_idc */
	bw_se(b, c->qp - 26); /* pic_init_qp_minus26 */
	bw_se(b, c->qp - 26); /* pic_init_qs_minus26 */
	bw_se(b, c->chroma_qp_offset);
	bw_put(b, 1, 1); /* deblocking_filter_control_present_flag */
	bw_put(b, c->intra_refresh > 0, 1); /* constrained_intra_pred_flag: refresh macroblocks must not lean on inter samples */
	bw_put(b, 0, 1); /* redundant_pic_cnt_present_flag */
	bw_trailing(b);
}

int h264o_write_sps(const h264o_config *cfg, uint8_t *out, int cap)
{
	uint8_t tmp[64];
	bitw b;

	bw_init(&b, tmp, sizeof(tmp));
	sps_rbsp(cfg, &b);
	return nal_write(out, cap, 3, 7, tmp, (int)(b.pos >> 3));
}

/* access_unit_delimiter_rbsp (H.264 7.3.2.4): primary_pic_type 0 = I slices only, 1 = I and P */
int h264o_write_aud(int idr, uint8_t *out, int cap)
{
	uint8_t tmp[4];
	bitw b;

	bw_init(&b, tmp, sizeof(tmp));
	bw_put(&b, idr ? 0 : 1, 3);
	bw_trailing(&b);
	return nal_write(out, cap, 0, 9, tmp, (int)(b.pos >> 3));
}

int h264o_write_pps(const h264o_config *cfg, uint8_t *out, int cap)
{
	uint8_t tmp[64];
	bitw b;

	bw_init(&b, tmp, sizeof(tmp));
	pps_rbsp(cfg, &b);
	return nal_write(out, cap, 3, 8, tmp, (int)(b.pos >> 3));
}

static void slice_header_bits_qp(const h264o_config *c, bitw *b, int idr,
				 int frame_num, int idr_pic_id, int first_mb, int qp_delta);

static void slice_header_bits(const h264o_config *c, bitw *b, int idr,
			      int frame_num, int idr_pic_id, int first_mb)
{
	slice_header_bits_qp(c, b, idr, frame_num, idr_pic_id, first_mb, 0);
}

static void slice_header_bits_qp(const h264o_config *c, bitw *b, int idr,
				 int frame_num, int idr_pic_id, int first_mb, int qp_delta)
{
	bw_ue(b, (uint32_t)first_mb);
	bw_ue(b, idr ? 2 : 0); /* slice_type I=2, P=0 (cedar.c:994-997) */
	bw_ue(b, 0);           /* pic_parameter_set_id */
	bw_put(b, (uint32_t)(frame_num & 0x0f), 4);
	if (idr) {
		bw_ue(b, (uint32_t)idr_pic_id);
		bw_put(b, 0, 1); /* no_output_of_prior_pics_flag */
		bw_put(b, 0, 1); /* long_term_reference_flag */
	} else {
		bw_put(b, 0, 1); /* num_ref_idx_active_override_flag */
		bw_put(b, 0, 1); /* ref_pic_list_modification_flag_l0 */
		bw_put(b, 0, 1); /* adaptive_ref_pic_marking_mode_flag */
	}
	bw_se(b, qp_delta); /* slice_qp_delta: 0 at fixed QP (cedar.c:1021-1022) */
	/* disable_deblocking_filter_idc: 0 single slice (cedar.c:1024-1025);
	 * 2 when the picture has several slices so they stay independent */
	bw_ue(b, c->slice_count > 1 ? 2 : 0);
	bw_se(b, 0); /* slice_alpha_c0_offset_div2 */
	bw_se(b, 0); /* slice_beta_offset_div2 */
}

int h264o_write_slice_header(const h264o_config *cfg, int idr, int frame_num,
			     int idr_pic_id, int first_mb, uint8_t *out,
			     int cap, int *nbits)
{
	uint8_t tmp[32];
	bitw b;

	bw_init(&b, tmp, sizeof(tmp));
	slice_header_bits(cfg, &b, idr, frame_num, idr_pic_id, first_mb);
	if (nbits)
		*nbits = (int)b.pos;
	return nal_write(out, cap, idr ? 3 : 2, idr ? 5 : 1, tmp,
			 (int)((b.pos + 7) >> 3));
}

/* ------------------------------------------------------------------ */
/* transforms and quantisation (clause 8.5)                            */

void h264o_fdct4x4(const int16_t in[16], int16_t out[16])
{
	int t[16], i;

	for (i = 0; i < 4; i++) {
		int x0 = in[4 * i], x1 = in[4 * i + 1], x2 = in[4 * i + 2], x3 = in[4 * i + 3];
		int a = x0 + x3, b = x1 + x2, c = x1 - x2, d = x0 - x3;

		t[4 * i] = a + b;
		t[4 * i + 1] = 2 * d + c;
		t[4 * i + 2] = a - b;
		t[4 * i + 3] = d - 2 * c;
	}
	for (i = 0; i < 4; i++) {
		int x0 = t[i], x1 = t[4 + i], x2 = t[8 + i], x3 = t[12 + i];
		int a = x0 + x3, b = x1 + x2, c = x1 - x2, d = x0 - x3;

		out[i] = (int16_t)(a + b);
		out[4 + i] = (int16_t)(2 * d + c);
		out[8 + i] = (int16_t)(a - b);
		out[12 + i] = (int16_t)(d - 2 * c);
	}
}

/* inverse core transform of scaled coefficients d[] (raster), rows first then
 * columns (clause 8.5.12.2), result (x+32)>>6 added to dst with clipping */
static void idct4x4_add_i32(const int d[16], uint8_t *dst, int stride)
{
	int t[16], i;

	for (i = 0; i < 4; i++) {
		int d0 = d[4 * i], d1 = d[4 * i + 1], d2 = d[4 * i + 2], d3 = d[4 * i + 3];
		int e0 = d0 + d2, e1 = d0 - d2, e2 = (d1 >> 1) - d3, e3 = d1 + (d3 >> 1);

		t[4 * i] = e0 + e3;
		t[4 * i + 1] = e1 + e2;
		t[4 * i + 2] = e1 - e2;
		t[4 * i + 3] = e0 - e3;
	}
	for (i = 0; i < 4; i++) {
		int d0 = t[i], d1 = t[4 + i], d2 = t[8 + i], d3 = t[12 + i];
		int e0 = d0 + d2, e1 = d0 - d2, e2 = (d1 >> 1) - d3, e3 = d1 + (d3 >> 1);
		int r[4], k;

		r[0] = e0 + e3;
		r[1] = e1 + e2;
		r[2] = e1 - e2;
		r[3] = e0 - e3;
		for (k = 0; k < 4; k++) {
			int v = dst[k * stride + i] + ((r[k] + 32) >> 6);

			dst[k * stride + i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
		}
	}
}

void h264o_idct4x4_add(const int16_t coef[16], uint8_t *dst, int stride)
{
	int d[16], i;

	for (i = 0; i < 16; i++)
		d[i] = coef[i];
	idct4x4_add_i32(d, dst, stride);
}

#define LEVEL_MAX 2047 /* keeps every level inside the baseline CAVLC escape range */

static int quant_one(int w, int mf, int f, int qbits)
{
	int a = w < 0 ? -w : w;
	int l = (int)(((long long)a * mf + f) >> qbits);

	if (l > LEVEL_MAX)
		l = LEVEL_MAX;
	return w < 0 ? -l : l;
}

/* ------------------------------------------------------------------ */
/* intra prediction (clause 8.3); one routine serves the open-loop     */
/* decision (neighbours from the source) and the closed-loop recon     */

typedef struct nb4 {
	int has_top, has_left, has_tr; /* top-left available iff top && left */
	uint8_t top[8];                /* top[4..7] = top-right (or replicated) */
	uint8_t left[4];
	uint8_t tl;
} nb4;

static void fetch_nb4(const uint8_t *pic, int stride, int x, int y,
		      int has_top, int has_left, int has_tr, nb4 *n)
{
	int i;

	n->has_top = has_top;
	n->has_left = has_left;
	n->has_tr = has_tr;
	memset(n->top, 128, 8);
	memset(n->left, 128, 4);
	n->tl = 128;
	if (has_top) {
		for (i = 0; i < 4; i++)
			n->top[i] = pic[(y - 1) * stride + x + i];
		for (i = 4; i < 8; i++)
			n->top[i] = has_tr ? pic[(y - 1) * stride + x + i] : n->top[3];
	}
	if (has_left)
		for (i = 0; i < 4; i++)
			n->left[i] = pic[(y + i) * stride + x - 1];
	if (has_top && has_left)
		n->tl = pic[(y - 1) * stride + x - 1];
}

static int mode4_allowed(int mode, const nb4 *n)
{
	switch (mode) {
	case 0: case 3: case 7:
		return n->has_top;
	case 1: case 8:
		return n->has_left;
	case 2:
		return 1;
	default: /* 4 DDR, 5 VR, 6 HD */
		return n->has_top && n->has_left;
	}
}

static void pred4x4(int mode, const nb4 *n, uint8_t p[16])
{
	/* e[] = the 13-sample edge: e[4+1+x] = top[x], e[4] = tl, e[3-y] = left[y] */
	const uint8_t *t = n->top, *l = n->left;
	int x, y;

	switch (mode) {
	case 0: /* vertical */
		for (y = 0; y < 4; y++)
			for (x = 0; x < 4; x++)
				p[4 * y + x] = t[x];
		break;
	case 1: /* horizontal */
		for (y = 0; y < 4; y++)
			for (x = 0; x < 4; x++)
				p[4 * y + x] = l[y];
		break;
	case 2: { /* DC */
		int s;

		if (n->has_top && n->has_left)
			s = (t[0] + t[1] + t[2] + t[3] + l[0] + l[1] + l[2] + l[3] + 4) >> 3;
		else if (n->has_left)
			s = (l[0] + l[1] + l[2] + l[3] + 2) >> 2;
		else if (n->has_top)
			s = (t[0] + t[1] + t[2] + t[3] + 2) >> 2;
		else
			s = 128;
		memset(p, s, 16);
		break;
	}
	case 3: /* diagonal down-left */
		for (y = 0; y < 4; y++)
			for (x = 0; x < 4; x++) {
				if (x == 3 && y == 3)
					p[4 * y + x] = (uint8_t)((t[6] + 3 * t[7] + 2) >> 2);
				else
					p[4 * y + x] = (uint8_t)((t[x + y] + 2 * t[x + y + 1] + t[x + y + 2] + 2) >> 2);
			}
		break;
	case 4: { /* diagonal down-right */
		uint8_t e[9];

		e[0] = l[3]; e[1] = l[2]; e[2] = l[1]; e[3] = l[0]; e[4] = n->tl;
		e[5] = t[0]; e[6] = t[1]; e[7] = t[2]; e[8] = t[3];
		for (y = 0; y < 4; y++)
			for (x = 0; x < 4; x++) {
				int k = 4 + x - y; /* centre sample index in e[] */

				p[4 * y + x] = (uint8_t)((e[k - 1] + 2 * e[k] + e[k + 1] + 2) >> 2);
			}
		break;
	}
	case 5: { /* vertical-right */
		uint8_t e[9];

		e[0] = l[3]; e[1] = l[2]; e[2] = l[1]; e[3] = l[0]; e[4] = n->tl;
		e[5] = t[0]; e[6] = t[1]; e[7] = t[2]; e[8] = t[3];
		for (y = 0; y < 4; y++)
			for (x = 0; x < 4; x++) {
				int z = 2 * x - y;

				if (z >= 0 && !(z & 1)) {
					int k = 4 + x - (y >> 1); /* p[x-(y>>1)-1,-1], p[x-(y>>1),-1] */

					p[4 * y + x] = (uint8_t)((e[k] + e[k + 1] + 1) >> 1);
				} else if (z >= 0) {
					int k = 4 + x - (y >> 1);

					p[4 * y + x] = (uint8_t)((e[k - 1] + 2 * e[k] + e[k + 1] + 2) >> 2);
				} else if (z == -1) {
					p[4 * y + x] = (uint8_t)((l[0] + 2 * n->tl + t[0] + 2) >> 2);
				} else {
					/* p[-1,y-1] + 2 p[-1,y-2] + p[-1,y-3] */
					int k = 4 - y; /* e[k]=left[y-1] */

					p[4 * y + x] = (uint8_t)((e[k] + 2 * e[k + 1] + e[k + 2] + 2) >> 2);
				}
			}
		break;
	}
	case 6: { /* horizontal-down */
		uint8_t e[9];

		e[0] = l[3]; e[1] = l[2]; e[2] = l[1]; e[3] = l[0]; e[4] = n->tl;
		e[5] = t[0]; e[6] = t[1]; e[7] = t[2]; e[8] = t[3];
		for (y = 0; y < 4; y++)
			for (x = 0; x < 4; x++) {
				int z = 2 * y - x;

				if (z >= 0 && !(z & 1)) {
					int k = 4 - (y - (x >> 1)); /* e[k]=p[-1,y-(x>>1)-1], e[k-1]=p[-1,y-(x>>1)] */

					p[4 * y + x] = (uint8_t)((e[k] + e[k - 1] + 1) >> 1);
				} else if (z >= 0) {
					int k = 4 - (y - (x >> 1));

					p[4 * y + x] = (uint8_t)((e[k + 1] + 2 * e[k] + e[k - 1] + 2) >> 2);
				} else if (z == -1) {
					p[4 * y + x] = (uint8_t)((l[0] + 2 * n->tl + t[0] + 2) >> 2);
				} else {
					/* p[x-1,-1] + 2 p[x-2,-1] + p[x-3,-1] */
					int k = 4 + x; /* e[k]=top[x-1] */

					p[4 * y + x] = (uint8_t)((e[k] + 2 * e[k - 1] + e[k - 2] + 2) >> 2);
				}
			}
		break;
	}
	case 7: /* vertical-left */
		for (y = 0; y < 4; y++)
			for (x = 0; x < 4; x++) {
				int k = x + (y >> 1);

				if (!(y & 1))
					p[4 * y + x] = (uint8_t)((t[k] + t[k + 1] + 1) >> 1);
				else
					p[4 * y + x] = (uint8_t)((t[k] + 2 * t[k + 1] + t[k + 2] + 2) >> 2);
			}
		break;
	default: /* 8 horizontal-up */
		for (y = 0; y < 4; y++)
			for (x = 0; x < 4; x++) {
				int z = x + 2 * y;

				if (z > 5)
					p[4 * y + x] = l[3];
				else if (z == 5)
					p[4 * y + x] = (uint8_t)((l[2] + 3 * l[3] + 2) >> 2);
				else if (!(z & 1)) {
					int k = y + (x >> 1);

					p[4 * y + x] = (uint8_t)((l[k] + l[k + 1] + 1) >> 1);
				} else {
					int k = y + (x >> 1);

					p[4 * y + x] = (uint8_t)((l[k] + 2 * l[k + 1] + l[k + 2] + 2) >> 2);
				}
			}
		break;
	}
}

static int clip255(int v)
{
	return v < 0 ? 0 : v > 255 ? 255 : v;
}

/* Intra 16x16 luma prediction (8.3.3); mode 0 V 1 H 2 DC 3 Plane */
static void pred16x16(int mode, const uint8_t *pic, int stride, int x0, int y0,
		      int has_top, int has_left, uint8_t p[256])
{
	int x, y;
	const uint8_t *top = has_top ? pic + (y0 - 1) * stride + x0 : pic;

	switch (mode) {
	case 0:
		for (y = 0; y < 16; y++)
			memcpy(p + 16 * y, top, 16);
		break;
	case 1:
		for (y = 0; y < 16; y++)
			memset(p + 16 * y, pic[(y0 + y) * stride + x0 - 1], 16);
		break;
	case 2: {
		int s = 0;

		if (has_top)
			for (x = 0; x < 16; x++)
				s += top[x];
		if (has_left)
			for (y = 0; y < 16; y++)
				s += pic[(y0 + y) * stride + x0 - 1];
		if (has_top && has_left)
			s = (s + 16) >> 5;
		else if (has_top || has_left)
			s = (s + 8) >> 4;
		else
			s = 128;
		memset(p, s, 256);
		break;
	}
	default: {
		int H = 0, V = 0, a, b, c;

		for (x = 0; x < 8; x++) {
			int hi = top[8 + x];
			int lo = x == 7 ? pic[(y0 - 1) * stride + x0 - 1] : top[6 - x];

			H += (x + 1) * (hi - lo);
		}
		for (y = 0; y < 8; y++) {
			int hi = pic[(y0 + 8 + y) * stride + x0 - 1];
			int lo = y == 7 ? pic[(y0 - 1) * stride + x0 - 1]
					: pic[(y0 + 6 - y) * stride + x0 - 1];

			V += (y + 1) * (hi - lo);
		}
		a = 16 * (pic[(y0 + 15) * stride + x0 - 1] + top[15]);
		b = (5 * H + 32) >> 6;
		c = (5 * V + 32) >> 6;
		for (y = 0; y < 16; y++)
			for (x = 0; x < 16; x++)
				p[16 * y + x] = (uint8_t)clip255((a + b * (x - 7) + c * (y - 7) + 16) >> 5);
		break;
	}
	}
}

/* Intra chroma prediction for one 8x8 plane stored interleaved (nv12): sample
 * (x,y) of plane c is uv[y*stride + 2*x + c]. mode 0 DC 1 H 2 V 3 Plane (8.3.4) */
static void pred_chroma8x8(int mode, const uint8_t *uv, int stride, int cx0, int cy0,
			   int c, int has_top, int has_left, uint8_t p[64])
{
	int x, y;
#define CPX(xx, yy) uv[((cy0) + (yy)) * stride + 2 * ((cx0) + (xx)) + c]

	switch (mode) {
	case 0: {
		int bx, by;

		for (by = 0; by < 2; by++)
			for (bx = 0; bx < 2; bx++) {
				int st = 0, sl = 0, v;

				if (has_top)
					for (x = 0; x < 4; x++)
						st += CPX(4 * bx + x, -1);
				if (has_left)
					for (y = 0; y < 4; y++)
						sl += CPX(-1, 4 * by + y);
				if (bx == by) { /* (0,0) and (1,1): both when both */
					if (has_top && has_left)
						v = (st + sl + 4) >> 3;
					else if (has_top)
						v = (st + 2) >> 2;
					else if (has_left)
						v = (sl + 2) >> 2;
					else
						v = 128;
				} else if (bx == 1) { /* (1,0): prefer top */
					if (has_top)
						v = (st + 2) >> 2;
					else if (has_left)
						v = (sl + 2) >> 2;
					else
						v = 128;
				} else { /* (0,1): prefer left */
					if (has_left)
						v = (sl + 2) >> 2;
					else if (has_top)
						v = (st + 2) >> 2;
					else
						v = 128;
				}
				for (y = 0; y < 4; y++)
					for (x = 0; x < 4; x++)
						p[8 * (4 * by + y) + 4 * bx + x] = (uint8_t)v;
			}
		break;
	}
	case 1:
		for (y = 0; y < 8; y++)
			for (x = 0; x < 8; x++)
				p[8 * y + x] = CPX(-1, y);
		break;
	case 2:
		for (y = 0; y < 8; y++)
			for (x = 0; x < 8; x++)
				p[8 * y + x] = CPX(x, -1);
		break;
	default: {
		int H = 0, V = 0, a, b, cc;

		for (x = 0; x < 4; x++)
			H += (x + 1) * (CPX(4 + x, -1) - CPX(2 - x, -1));
		for (y = 0; y < 4; y++)
			V += (y + 1) * (CPX(-1, 4 + y) - CPX(-1, 2 - y));
		a = 16 * (CPX(-1, 7) + CPX(7, -1));
		b = (34 * H + 32) >> 6;
		cc = (34 * V + 32) >> 6;
		for (y = 0; y < 8; y++)
			for (x = 0; x < 8; x++)
				p[8 * y + x] = (uint8_t)clip255((a + b * (x - 3) + cc * (y - 3) + 16) >> 5);
		break;
	}
	}
#undef CPX
}

/* ------------------------------------------------------------------ */
/* geometry helpers                                                    */

static void slice_rows(const h264o_enc *e, int s, int *r0, int *r1)
{
	int n = e->cfg.slice_count, base = e->mb_h / n, rem = e->mb_h % n;
	int start = s * base + (s < rem ? s : rem);

	*r0 = start;
	*r1 = start + base + (s < rem ? 1 : 0);
}

static int slice_of_row(const h264o_enc *e, int row)
{
	int s, r0, r1;

	for (s = 0; s < e->cfg.slice_count; s++) {
		slice_rows(e, s, &r0, &r1);
		if (row >= r0 && row < r1)
			return s;
	}
	return -1;
}

/* internal top-right availability by coding-order index for blocks not in
 * the top row of the macroblock (decoding-order rule, clause 6.4.11.4) */
static const uint8_t tr_internal[16] = { 0, 0, 1, 0, 0, 0, 1, 0, 1, 1, 1, 0, 1, 0, 1, 0 };

static void blk4_avail(const h264o_enc *e, int mx, int my, int row0, int idx,
		       int *has_top, int *has_left, int *has_tr)
{
	int bx = h264o_blk_x[idx], by = h264o_blk_y[idx];
	int top_mb = my > row0, left_mb = mx > 0;
	int tr_mb = top_mb && mx < e->mb_w - 1;

	*has_top = by > 0 || top_mb;
	*has_left = bx > 0 || left_mb;
	if (by == 0)
		*has_tr = bx < 3 ? top_mb : tr_mb;
	else
		*has_tr = tr_internal[idx];
}

static int chroma_qp_of(const h264o_enc *e)
{
	int q = e->mbqp + e->cfg.chroma_qp_offset;

	return h264o_chroma_qp[q < 0 ? 0 : q > 51 ? 51 : q];
}

/* ------------------------------------------------------------------ */
/* shared residual coding of chroma for intra and inter macroblocks    */

/* MAD classes (cfg.mad_th / mad_dqp): the macroblock's mean absolute difference against the prediction its
 * decision chose -- luma SAD / 256 -- falls into one of five classes, each with a QP offset on top of the
 * picture's QP and the region's. The decision itself (lambda) is taken before, at the QP without it. */
/* Sessions whose macroblocks differ in QP: the first macroblock of every row always carries an mb_qp_delta.
 * QP_Y of a macroblock without one is its predecessor's in decoding order -- across a row boundary the LAST
 * macroblock of the row above, far away on the analysis / deblock diagonals x + 2y. With the row's first
 * macroblock "coded" the chain never leaves the row, and deblocking can follow it left to right. A macroblock
 * that has no coded block gets coded_block_pattern = "chroma DC", two empty chroma DC blocks (4 bits) and its
 * mb_qp_delta; a would-be P_Skip becomes P_L0_16x16 with that. */
static void force_row_start_coded(const h264o_enc *e, h264o_mb *mb, int mx)
{
	if ((e->cfg.roi_count > 0 || e->mad_on) && mx == 0 && mb->type != H264O_MB_I16x16 && !(mb->cbp & 0x3f))
		mb->cbp = 0x10;
}

static void mad_qp(h264o_enc *e, int mx, int my, int sad)
{
	const int *th = e->cfg.mad_th, mad = sad >> 8;
	const int cls = (mad >= th[0]) + (mad >= th[1]) + (mad >= th[2]) + (mad >= th[3]);
	int q;

	if (!e->mad_on)
		return;
	q = e->qp_t[my * e->mb_w + mx] + e->cfg.mad_dqp[cls];
	e->qp_t[my * e->mb_w + mx] = (uint8_t)(q < 1 ? 1 : q > 51 ? 51 : q);
	e->mbqp = e->qp_t[my * e->mb_w + mx];
}

static void chroma_residual(h264o_enc *e, h264o_mb *mb, int mx, int my,
			    const uint8_t predc[2][64], int intra)
{
	int qpc = chroma_qp_of(e);
	int qbits = 15 + qpc / 6, rem = qpc % 6, qs = qpc / 6;
	int f = (1 << qbits) / (intra ? 3 : 6);
	uint8_t *rec_uv = e->rec_uv[e->cur];
	int stride = e->cw;
	int c, b, i, any_ac = 0, any_dc = 0;
	int16_t w[2][4][16];

	for (c = 0; c < 2; c++) {
		int dc[4], hd[4], dcr[4];

		for (b = 0; b < 4; b++) {
			int bx = b & 1, by = b >> 1;
			int16_t res[16];

			for (i = 0; i < 16; i++) {
				int x = 4 * bx + (i & 3), y = 4 * by + (i >> 2);
				int s = e->src_uv[(my * 8 + y) * stride + 2 * (mx * 8 + x) + c];

				res[i] = (int16_t)(s - predc[c][8 * y + x]);
			}
			h264o_fdct4x4(res, w[c][b]);
			dc[b] = w[c][b][0];
		}
		/* 2x2 Hadamard of the DCs, raster order */
		hd[0] = dc[0] + dc[1] + dc[2] + dc[3];
		hd[1] = dc[0] - dc[1] + dc[2] - dc[3];
		hd[2] = dc[0] + dc[1] - dc[2] - dc[3];
		hd[3] = dc[0] - dc[1] - dc[2] + dc[3];
		for (b = 0; b < 4; b++) {
			mb->cdc[c][b] = (int16_t)quant_one(hd[b], h264o_quant_mf[rem][0], 2 * f, qbits + 1);
			if (mb->cdc[c][b])
				any_dc = 1;
		}
		for (b = 0; b < 4; b++) {
			int n = 0;

			mb->cac[c][b][0] = 0;
			for (i = 1; i < 16; i++) {
				int r = h264o_zigzag4x4[i];
				int l = quant_one(w[c][b][r], h264o_quant_mf[rem][h264o_pos_class[r]], f, qbits);

				mb->cac[c][b][i] = (int16_t)l;
				if (l)
					n++;
			}
			mb->nnz[16 + 4 * c + b] = (uint8_t)n;
			if (n)
				any_ac = 1;
		}
		/* reconstruct: inverse 2x2 Hadamard, DC scaling (8.5.11.2) */
		{
			int c0 = mb->cdc[c][0], c1 = mb->cdc[c][1], c2 = mb->cdc[c][2], c3 = mb->cdc[c][3];
			int fd[4];

			fd[0] = c0 + c1 + c2 + c3;
			fd[1] = c0 - c1 + c2 - c3;
			fd[2] = c0 + c1 - c2 - c3;
			fd[3] = c0 - c1 - c2 + c3;
			for (b = 0; b < 4; b++)
				dcr[b] = ((fd[b] * h264o_dequant_v[rem][0] * 16) * (1 << qs)) >> 5;
		}
		for (b = 0; b < 4; b++) {
			int bx = b & 1, by = b >> 1;
			int d[16];
			uint8_t blk[16];

			d[0] = dcr[b];
			for (i = 1; i < 16; i++) {
				int r = h264o_zigzag4x4[i];

				d[r] = (mb->cac[c][b][i] * h264o_dequant_v[rem][h264o_pos_class[r]]) * (1 << qs);
			}
			for (i = 0; i < 16; i++)
				blk[i] = predc[c][8 * (4 * by + (i >> 2)) + 4 * bx + (i & 3)];
			idct4x4_add_i32(d, blk, 4);
			for (i = 0; i < 16; i++) {
				int x = 4 * bx + (i & 3), y = 4 * by + (i >> 2);

				rec_uv[(my * 8 + y) * stride + 2 * (mx * 8 + x) + c] = blk[i];
			}
		}
	}
	mb->cbp = (int8_t)((mb->cbp & 15) | ((any_ac ? 2 : any_dc ? 1 : 0) << 4));
}

/* luma 4x4 block: residual -> levels (zig-zag) -> reconstruction at rec */
static int luma_block_16(const h264o_enc *e, const int16_t res[16], int16_t lv[16],
			 uint8_t *rec, int stride, int intra)
{
	int qp = e->mbqp, qbits = 15 + qp / 6, rem = qp % 6, qs = qp / 6;
	int f = (1 << qbits) / (intra ? 3 : 6);
	int16_t w[16];
	int d[16], i, n = 0;

	h264o_fdct4x4(res, w);
	for (i = 0; i < 16; i++) {
		int r = h264o_zigzag4x4[i];
		int l = quant_one(w[r], h264o_quant_mf[rem][h264o_pos_class[r]], f, qbits);

		lv[i] = (int16_t)l;
		if (l)
			n++;
		d[r] = (l * h264o_dequant_v[rem][h264o_pos_class[r]]) * (1 << qs);
	}
	idct4x4_add_i32(d, rec, stride); /* rec holds the prediction on entry */
	return n;
}

/* ------------------------------------------------------------------ */
/* I_PCM (cfg.pcm): H.264 A.3.1 allows macroblock_layer() 3200 bits.    */

/* What CAVLC spends on a macroblock, from its levels alone: every block's residual_block_cavlc() in full --
 * level codes with their suffixLength, total_zeros, run_before -- with coeff_token at the longest of the four
 * tables (which one applies depends on the neighbours' counts); 100 bits cover mb_type, prediction modes or
 * vector differences, coded_block_pattern and mb_qp_delta (they depend on neighbours too). */
static void cavlc_block(bitw *b, const int16_t *lv, int maxc, int nC);

static int pcm_block_bits(const int16_t *lv, int maxc, int chroma_dc)
{
	static const int ncs[4] = { 0, 2, 4, 8 };
	int best = 0, k;

	for (k = 0; k < (chroma_dc ? 1 : 4); k++) {
		uint8_t tmp[128];
		bitw b;

		bw_init(&b, tmp, sizeof(tmp));
		cavlc_block(&b, lv, maxc, chroma_dc ? -1 : ncs[k]);
		if ((int)b.pos > best)
			best = (int)b.pos;
	}
	return best;
}

#define PCM_HEADER_BITS 100
#define PCM_LIMIT 3200      /* P pictures: the levels are final, the count holds for what is written */
#define PCM_LIMIT_OPEN 2900 /* I pictures: the count is taken on the open loop's levels (below) */

/* The sessions and pictures that take I_PCM. One QP per picture (an I_PCM macroblock carries no mb_qp_delta and
 * leaves QP_Y as it finds it, 7.4.5). Only the deblocking filter takes its QP as 0 (8.7.2.2): its edges are filtered
 * at the mean of the two QP_Y resp. QP_C, and the pictures taken are those where that
 * mean has alpha = 0 on both planes, i.e. nothing is filtered there at all: QP <= 30 for luma, and QP_C(QP) +
 * QP_C(0) + 1 <= 31 for chroma -- QP <= 22 with the reference's chroma offset of 4. (Noise passes 3200 bits up to
 * QP 16; above the bound strict_mb_bits still says what happens.) No refresh macroblocks: under
 * constrained_intra_pred they would have to predict from an I_PCM neighbour. */
static int chroma_qp_at(const h264o_enc *e, int qp_y);

static int pcm_allowed(const h264o_enc *e)
{
	if (e->cfg.pcm < 0 || e->cfg.roi_count > 0 || e->mad_on || e->cfg.intra_refresh > 0)
		return 0;
	return h264o_alpha[(e->qp + 1) >> 1] == 0 && h264o_alpha[(chroma_qp_at(e, e->qp) + chroma_qp_at(e, 0) + 1) >> 1] == 0;
}

static void pcm_make(h264o_enc *e, int mx, int my)
{
	h264o_mb *mb = &e->mbs[my * e->mb_w + mx];
	int y;

	mb->type = H264O_MB_IPCM;
	mb->i16_mode = 0;
	mb->chroma_mode = 0;
	mb->cbp = 0;
	mb->mvx = mb->mvy = 0;
	memset(mb->i4_modes, 2, sizeof(mb->i4_modes));
	memset(mb->nnz, 16, sizeof(mb->nnz)); /* 9.2.1: nC next to an I_PCM macroblock counts 16 */
	memset(mb->luma, 0, sizeof(mb->luma));
	memset(mb->dc16, 0, sizeof(mb->dc16));
	memset(mb->cdc, 0, sizeof(mb->cdc));
	memset(mb->cac, 0, sizeof(mb->cac));
	for (y = 0; y < 16; y++)
		memcpy(e->rec_y[e->cur] + (size_t)(my * 16 + y) * e->cw + mx * 16, e->src_y + (size_t)(my * 16 + y) * e->cw + mx * 16, 16);
	for (y = 0; y < 8; y++)
		memcpy(e->rec_uv[e->cur] + (size_t)(my * 8 + y) * e->cw + mx * 16, e->src_uv + (size_t)(my * 8 + y) * e->cw + mx * 16, 16);
}

/* ------------------------------------------------------------------ */
/* I pictures                                                          */

static int sad_block(const uint8_t *a, int sa, const uint8_t *b, int sb, int w, int h)
{
	int x, y, s = 0;

	for (y = 0; y < h; y++)
		for (x = 0; x < w; x++) {
			int d = a[y * sa + x] - b[y * sb + x];

			s += d < 0 ? -d : d;
		}
	return s;
}

/* open-loop decision from source samples only: every macroblock independent */
static void intra_decide_mb(h264o_enc *e, int mx, int my, int row0)
{
	h264o_mb *mb = &e->mbs[my * e->mb_w + mx];
	/* (closed_loop: the planes the decision reads are the reconstruction with this macroblock's
	 * source samples laid in, see h264o_encode -- a measurement of what the open loop costs) */
	const uint8_t *sy = e->closed_loop ? e->rec_y[e->cur] : e->src_y, *suv = e->closed_loop ? e->rec_uv[e->cur] : e->src_uv;
	int stride = e->cw, x0 = mx * 16, y0 = my * 16;
	int top_mb = my > row0, left_mb = mx > 0;
	int idx, mode, sum4 = 0, best16 = -1, best16_sad = 0;
	int bestc = 0, bestc_sad = 0;

	for (idx = 0; idx < 16; idx++) {
		int bx = h264o_blk_x[idx], by = h264o_blk_y[idx];
		int ht, hl, htr, best = -1, best_sad = 0;
		nb4 n;

		blk4_avail(e, mx, my, row0, idx, &ht, &hl, &htr);
		fetch_nb4(sy, stride, x0 + 4 * bx, y0 + 4 * by, ht, hl, htr, &n);
		for (mode = 0; mode < 9; mode++) {
			uint8_t p[16];
			int s;

			if (!mode4_allowed(mode, &n))
				continue;
			pred4x4(mode, &n, p);
			s = sad_block(sy + (y0 + 4 * by) * stride + x0 + 4 * bx, stride, p, 4, 4, 4);
			if (best < 0 || s < best_sad) {
				best = mode;
				best_sad = s;
			}
		}
		mb->i4_modes[idx] = (int8_t)best;
		sum4 += best_sad;
	}
	for (mode = 0; mode < 4; mode++) {
		uint8_t p[256];
		int s;

		if ((mode == 0 && !top_mb) || (mode == 1 && !left_mb) ||
		    (mode == 3 && !(top_mb && left_mb)))
			continue;
		pred16x16(mode, sy, stride, x0, y0, top_mb, left_mb, p);
		s = sad_block(sy + y0 * stride + x0, stride, p, 16, 16, 16);
		if (best16 < 0 || s < best16_sad) {
			best16 = mode;
			best16_sad = s;
		}
	}
	mb->i16_mode = (int8_t)best16;
	mb->type = (sum4 + 32 * h264o_lambda[e->mbqp] < best16_sad) ? H264O_MB_I4x4 : H264O_MB_I16x16;
	mad_qp(e, mx, my, mb->type == H264O_MB_I4x4 ? sum4 : best16_sad);

	bestc = -1;
	for (mode = 0; mode < 4; mode++) {
		int c, s = 0;

		if ((mode == 1 && !left_mb) || (mode == 2 && !top_mb) ||
		    (mode == 3 && !(top_mb && left_mb)))
			continue;
		for (c = 0; c < 2; c++) {
			uint8_t p[64];
			int x, y;

			pred_chroma8x8(mode, suv, stride, mx * 8, my * 8, c, top_mb, left_mb, p);
			for (y = 0; y < 8; y++)
				for (x = 0; x < 8; x++) {
					int d = suv[(my * 8 + y) * stride + 2 * (mx * 8 + x) + c] - p[8 * y + x];

					s += d < 0 ? -d : d;
				}
		}
		if (bestc < 0 || s < bestc_sad) {
			bestc = mode;
			bestc_sad = s;
		}
	}
	mb->chroma_mode = (int8_t)bestc;
	mb->mvx = mb->mvy = 0;

	/* I_PCM, decided open loop like everything here: the residual of the CHOSEN predictions from source samples,
	 * every 4x4 block -- 16 luma, 4 + 4 chroma -- through the plain transform and the intra quantiser (no DC
	 * Hadamards), its zig-zag levels through the count above. The closed loop's levels differ a little: the limit
	 * here leaves a tenth of room (and strict_mb_bits still looks at what was really written). */
	if (pcm_allowed(e)) {
		int bits = PCM_HEADER_BITS, qp = e->mbqp, qpc = chroma_qp_of(e), c;

		for (idx = 0; idx < 16; idx++) {
			int bx = h264o_blk_x[idx], by = h264o_blk_y[idx], x, y, i;
			uint8_t p4[16], p16[256];
			int16_t res[16], w[16], lv[16];

			if (mb->type == H264O_MB_I4x4) {
				int ht, hl, htr;
				nb4 n;

				blk4_avail(e, mx, my, row0, idx, &ht, &hl, &htr);
				fetch_nb4(sy, stride, x0 + 4 * bx, y0 + 4 * by, ht, hl, htr, &n);
				pred4x4(mb->i4_modes[idx], &n, p4);
			} else {
				pred16x16(mb->i16_mode, sy, stride, x0, y0, top_mb, left_mb, p16);
				for (y = 0; y < 4; y++)
					for (x = 0; x < 4; x++)
						p4[4 * y + x] = p16[16 * (4 * by + y) + 4 * bx + x];
			}
			for (y = 0; y < 4; y++)
				for (x = 0; x < 4; x++)
					res[4 * y + x] = (int16_t)(sy[(y0 + 4 * by + y) * stride + x0 + 4 * bx + x] - p4[4 * y + x]);
			h264o_fdct4x4(res, w);
			for (i = 0; i < 16; i++)
				lv[i] = (int16_t)quant_one(w[h264o_zigzag4x4[i]], h264o_quant_mf[qp % 6][h264o_pos_class[h264o_zigzag4x4[i]]], (1 << (15 + qp / 6)) / 3, 15 + qp / 6);
			bits += pcm_block_bits(lv, 16, 0);
		}
		for (c = 0; c < 2; c++) {
			uint8_t pc[64];
			int b;

			pred_chroma8x8(mb->chroma_mode, suv, stride, mx * 8, my * 8, c, top_mb, left_mb, pc);
			for (b = 0; b < 4; b++) {
				int bx = b & 1, by = b >> 1, x, y, i;
				int16_t res[16], w[16], lv[16];

				for (y = 0; y < 4; y++)
					for (x = 0; x < 4; x++)
						res[4 * y + x] = (int16_t)(suv[(my * 8 + 4 * by + y) * stride + 2 * (mx * 8 + 4 * bx + x) + c] - pc[8 * (4 * by + y) + 4 * bx + x]);
				h264o_fdct4x4(res, w);
				for (i = 0; i < 16; i++)
					lv[i] = (int16_t)quant_one(w[h264o_zigzag4x4[i]], h264o_quant_mf[qpc % 6][h264o_pos_class[h264o_zigzag4x4[i]]], (1 << (15 + qpc / 6)) / 3, 15 + qpc / 6);
				bits += pcm_block_bits(lv, 16, 0);
			}
		}
		if (bits > PCM_LIMIT_OPEN)
			mb->type = H264O_MB_IPCM;
	}
}

/* isolated: a refresh macroblock of a P picture -- under constrained_intra_pred its (inter) neighbours
 * do not exist for intra prediction */
static void intra_recon_mb(h264o_enc *e, int mx, int my, int row0, int isolated)
{
	h264o_mb *mb = &e->mbs[my * e->mb_w + mx];
	uint8_t *ry = e->rec_y[e->cur], *ruv = e->rec_uv[e->cur];
	int stride = e->cw, x0 = mx * 16, y0 = my * 16;
	int top_mb = !isolated && my > row0, left_mb = !isolated && mx > 0;
	int qp = e->mbqp, qbits = 15 + qp / 6, rem = qp % 6, qs = qp / 6;
	int idx, i, c;
	uint8_t predc[2][64];

	if (mb->type == H264O_MB_IPCM) { /* its reconstruction is its source: what the neighbours predict from */
		pcm_make(e, mx, my);
		return;
	}
	memset(mb->nnz, 0, sizeof(mb->nnz));
	memset(mb->luma, 0, sizeof(mb->luma));
	memset(mb->dc16, 0, sizeof(mb->dc16));
	mb->cbp = 0;

	if (mb->type == H264O_MB_I4x4) {
		for (idx = 0; idx < 16; idx++) {
			int bx = h264o_blk_x[idx], by = h264o_blk_y[idx];
			int ht, hl, htr, n, x, y;
			uint8_t p[16];
			int16_t res[16];
			uint8_t *dst = ry + (y0 + 4 * by) * stride + x0 + 4 * bx;
			nb4 nb;

			blk4_avail(e, mx, my, row0, idx, &ht, &hl, &htr);
			fetch_nb4(ry, stride, x0 + 4 * bx, y0 + 4 * by, ht, hl, htr, &nb);
			pred4x4(mb->i4_modes[idx], &nb, p);
			for (y = 0; y < 4; y++)
				for (x = 0; x < 4; x++) {
					res[4 * y + x] = (int16_t)(e->src_y[(y0 + 4 * by + y) * stride + x0 + 4 * bx + x] - p[4 * y + x]);
					dst[y * stride + x] = p[4 * y + x];
				}
			n = luma_block_16(e, res, mb->luma[idx], dst, stride, 1);
			mb->nnz[bx + 4 * by] = (uint8_t)n;
			if (n)
				mb->cbp |= (int8_t)(1 << (idx >> 2));
		}
	} else {
		uint8_t p[256];
		int16_t w[16][16];
		int dcm[16], hd[16], dcr[16], any_ac = 0;
		int f = (1 << qbits) / 3;

		pred16x16(mb->i16_mode, ry, stride, x0, y0, top_mb, left_mb, p);
		for (idx = 0; idx < 16; idx++) {
			int bx = h264o_blk_x[idx], by = h264o_blk_y[idx], n = 0;
			int16_t res[16];

			for (i = 0; i < 16; i++) {
				int x = 4 * bx + (i & 3), y = 4 * by + (i >> 2);

				res[i] = (int16_t)(e->src_y[(y0 + y) * stride + x0 + x] - p[16 * y + x]);
			}
			h264o_fdct4x4(res, w[idx]);
			dcm[bx + 4 * by] = w[idx][0];
			mb->luma[idx][0] = 0;
			for (i = 1; i < 16; i++) {
				int r = h264o_zigzag4x4[i];
				int l = quant_one(w[idx][r], h264o_quant_mf[rem][h264o_pos_class[r]], f, qbits);

				mb->luma[idx][i] = (int16_t)l;
				if (l)
					n++;
			}
			mb->nnz[bx + 4 * by] = (uint8_t)n;
			if (n)
				any_ac = 1;
		}
		mb->cbp = (int8_t)(any_ac ? 15 : 0);
		/* forward 4x4 Hadamard of the DCs, (x+1)>>1, quantise with 2f, qbits+1 */
		{
			int t[16];

			for (i = 0; i < 4; i++) {
				int a = dcm[4 * i] + dcm[4 * i + 3], b = dcm[4 * i + 1] + dcm[4 * i + 2];
				int cc = dcm[4 * i + 1] - dcm[4 * i + 2], d = dcm[4 * i] - dcm[4 * i + 3];

				t[4 * i] = a + b;
				t[4 * i + 1] = d + cc;
				t[4 * i + 2] = a - b;
				t[4 * i + 3] = d - cc;
			}
			for (i = 0; i < 4; i++) {
				int a = t[i] + t[12 + i], b = t[4 + i] + t[8 + i];
				int cc = t[4 + i] - t[8 + i], d = t[i] - t[12 + i];

				hd[i] = (a + b + 1) >> 1;
				hd[4 + i] = (d + cc + 1) >> 1;
				hd[8 + i] = (a - b + 1) >> 1;
				hd[12 + i] = (d - cc + 1) >> 1;
			}
			for (i = 0; i < 16; i++)
				mb->dc16[i] = (int16_t)quant_one(hd[h264o_zigzag4x4[i]],
						h264o_quant_mf[rem][0], 2 * f, qbits + 1);
		}
		/* inverse Hadamard + DC scaling (8.5.10) */
		{
			int cm[16], t[16];

			for (i = 0; i < 16; i++)
				cm[h264o_zigzag4x4[i]] = mb->dc16[i];
			for (i = 0; i < 4; i++) {
				int a = cm[4 * i] + cm[4 * i + 3], b = cm[4 * i + 1] + cm[4 * i + 2];
				int cc = cm[4 * i + 1] - cm[4 * i + 2], d = cm[4 * i] - cm[4 * i + 3];

				t[4 * i] = a + b;
				t[4 * i + 1] = d + cc;
				t[4 * i + 2] = a - b;
				t[4 * i + 3] = d - cc;
			}
			for (i = 0; i < 4; i++) {
				int a = t[i] + t[12 + i], b = t[4 + i] + t[8 + i];
				int cc = t[4 + i] - t[8 + i], d = t[i] - t[12 + i];
				int fv[4], k;

				fv[0] = a + b;
				fv[1] = d + cc;
				fv[2] = a - b;
				fv[3] = d - cc;
				for (k = 0; k < 4; k++) {
					int ls = 16 * h264o_dequant_v[rem][0];

					if (qp >= 36)
						dcr[4 * k + i] = (fv[k] * ls) * (1 << (qs - 6));
					else
						dcr[4 * k + i] = (fv[k] * ls + (1 << (5 - qs))) >> (6 - qs);
				}
			}
		}
		for (idx = 0; idx < 16; idx++) {
			int bx = h264o_blk_x[idx], by = h264o_blk_y[idx], x, y;
			int d[16];
			uint8_t *dst = ry + (y0 + 4 * by) * stride + x0 + 4 * bx;

			d[0] = dcr[bx + 4 * by];
			for (i = 1; i < 16; i++) {
				int r = h264o_zigzag4x4[i];

				d[r] = (mb->luma[idx][i] * h264o_dequant_v[rem][h264o_pos_class[r]]) * (1 << qs);
			}
			for (y = 0; y < 4; y++)
				for (x = 0; x < 4; x++)
					dst[y * stride + x] = p[16 * (4 * by + y) + 4 * bx + x];
			idct4x4_add_i32(d, dst, stride);
		}
	}
	for (c = 0; c < 2; c++)
		pred_chroma8x8(mb->chroma_mode, ruv, stride, mx * 8, my * 8, c, top_mb, left_mb, predc[c]);
	chroma_residual(e, mb, mx, my, predc, 1);
}

/* ------------------------------------------------------------------ */
/* P pictures                                                          */

static int mv_bits(int d)
{
	int a = d < 0 ? -d : d;

	if (!a)
		return 1;
	return 2 * (bitlen32((uint32_t)(8 * a)) - 1) + 1;
}

/* ---- fractional-sample luma prediction, clause 8.4.2.2.1 (the caller keeps every tap inside the
 * picture, so no clamping here) ---- */
static int tap6(int a, int b, int c, int d, int e_, int f)
{
	return a - 5 * b + 20 * c + 20 * d - 5 * e_ + f;
}

static int clip8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

static int luma_sample_q(const uint8_t *r, int stride, int x, int y, int fx, int fy)
{
#define G_(dx, dy) ((int)r[(y + (dy)) * stride + x + (dx)])
#define B1(dx, dy) tap6(G_((dx) - 2, dy), G_((dx) - 1, dy), G_(dx, dy), G_((dx) + 1, dy), G_((dx) + 2, dy), G_((dx) + 3, dy))
#define H1(dx, dy) tap6(G_(dx, (dy) - 2), G_(dx, (dy) - 1), G_(dx, dy), G_(dx, (dy) + 1), G_(dx, (dy) + 2), G_(dx, (dy) + 3))
	int g = G_(0, 0), b, h, s, m, j;

	if (!fx && !fy)
		return g;
	if (!fy) {
		b = clip8((B1(0, 0) + 16) >> 5);
		return fx == 2 ? b : fx == 1 ? (g + b + 1) >> 1 : (G_(1, 0) + b + 1) >> 1;
	}
	if (!fx) {
		h = clip8((H1(0, 0) + 16) >> 5);
		return fy == 2 ? h : fy == 1 ? (g + h + 1) >> 1 : (G_(0, 1) + h + 1) >> 1;
	}
	b = clip8((B1(0, 0) + 16) >> 5);
	h = clip8((H1(0, 0) + 16) >> 5);
	s = clip8((B1(0, 1) + 16) >> 5);
	m = clip8((H1(1, 0) + 16) >> 5);
	j = clip8((tap6(B1(0, -2), B1(0, -1), B1(0, 0), B1(0, 1), B1(0, 2), B1(0, 3)) + 512) >> 10);
	if (fx == 2 && fy == 2)
		return j;
	if (fx == 2)
		return fy == 1 ? (b + j + 1) >> 1 : (s + j + 1) >> 1;
	if (fy == 2)
		return fx == 1 ? (h + j + 1) >> 1 : (m + j + 1) >> 1;
	return ((fy == 1 ? b : s) + (fx == 1 ? h : m) + 1) >> 1;
#undef G_
#undef B1
#undef H1
}

/* bits of se(v) for a vector component in quarter samples (== mv_bits(d) for v = 4 d) */
static int mv_bits_q(int v)
{
	int a = v < 0 ? -v : v, n = 0;

	if (!a)
		return 1;
	for (a *= 2; a > 1; a >>= 1)
		n++;
	return 2 * n + 1;
}

/* May the 16x16 block at quarter-sample vector (qx, qy) be predicted from?  Every sample a tap reads
 * lies inside the picture and the slice's band of rows (slices stay independent of each other), and
 * inside the +-16 window around the macroblock that the search works in. */
static int subpel_ok(const h264o_enc *e, int x0, int y0, int row0, int row1, int qx, int qy)
{
	int R4 = 4 * e->cfg.search_range;
	int ix = qx >> 2, iy = qy >> 2, fx = qx & 3, fy = qy & 3;
	int X = x0 + ix, Y = y0 + iy;

	if (qx < -R4 || qx > R4 || qy < -R4 || qy > R4)
		return 0;
	if (X < 0 || X + 16 > e->cw || Y < row0 * 16 || Y + 16 > row1 * 16)
		return 0;
	if (fx && (X - 2 < 0 || X + 18 >= e->cw || ix < -14 || ix > 13))
		return 0;
	if (fy && (Y - 2 < row0 * 16 || Y + 18 >= row1 * 16 || iy < -14 || iy > 13))
		return 0;
	return 1;
}

static void inter_mb(h264o_enc *e, int mx, int my, int row0, int row1)
{
	h264o_mb *mb = &e->mbs[my * e->mb_w + mx];
	const uint8_t *refy = e->rec_y[e->cur ^ 1], *refuv = e->rec_uv[e->cur ^ 1];
	uint8_t *ry = e->rec_y[e->cur];
	int stride = e->cw, x0 = mx * 16, y0 = my * 16;
	int R = e->cfg.search_range, lam = h264o_lambda[e->mbqp];
	int dx, dy, idx, c, qx, qy;
	long best_key = -1, best_of_refine = 0;
	int bdx = 0, bdy = 0;
	uint8_t predc[2][64];

	for (dy = -R; dy <= R; dy++) {
		if (y0 + dy < row0 * 16 || y0 + dy + 16 > row1 * 16)
			continue;
		for (dx = -R; dx <= R; dx++) {
			int s;
			long key;

			if (x0 + dx < 0 || x0 + dx + 16 > e->cw)
				continue;
			s = sad_block(e->src_y + y0 * stride + x0, stride,
				      refy + (y0 + dy) * stride + x0 + dx, stride, 16, 16);
			s += lam * (mv_bits(dx) + mv_bits(dy));
			key = ((long)s << 11) | ((dy + 16) * SEARCH_GRID + (dx + 16));
			if (best_key < 0 || key < best_key) {
				best_key = key;
				bdx = dx;
				bdy = dy;
			}
		}
	}
	qx = 4 * bdx;
	qy = 4 * bdy;
	/* Sub-sample refinement (cfg.subpel): the 8 half-sample neighbours of the best whole-sample vector,
	 * then the 8 quarter-sample neighbours of the winner. Cost as above, SAD of the interpolated
	 * prediction + lambda * bits of the vector; a candidate replaces the incumbent only when strictly
	 * cheaper, candidates in raster order (dy, then dx). */
	{
		long best = best_key >> 11; /* SAD + lambda * bits of the incumbent */
		int stage;

		for (stage = 1; stage <= e->cfg.subpel && stage <= 2; stage++) {
			int step = stage == 1 ? 2 : 1, cx = qx, cy = qy, sy, sx;

			for (sy = -step; sy <= step; sy += step)
				for (sx = -step; sx <= step; sx += step) {
					int vx = cx + sx, vy = cy + sy, x, y;
					long cost;

					if ((!sx && !sy) || !subpel_ok(e, x0, y0, row0, row1, vx, vy))
						continue;
					cost = lam * (mv_bits_q(vx) + mv_bits_q(vy));
					for (y = 0; y < 16; y++)
						for (x = 0; x < 16; x++) {
							int d = e->src_y[(y0 + y) * stride + x0 + x] -
								luma_sample_q(refy, stride, x0 + (vx >> 2) + x, y0 + (vy >> 2) + y, vx & 3, vy & 3);

							cost += d < 0 ? -d : d;
						}
					if (cost < best) {
						best = cost;
						qx = vx;
						qy = vy;
					}
				}
		}
		best_of_refine = best;
	}
	/* scene cuts (cfg.scene_cut): does the macroblock's own mean predict it better than the best vector? */
	if (e->cfg.scene_cut > 0) {
		int sum = 0, sadm = 0, mean, x, y;

		for (y = 0; y < 16; y++)
			for (x = 0; x < 16; x++)
				sum += e->src_y[(y0 + y) * stride + x0 + x];
		mean = (sum + 128) >> 8;
		for (y = 0; y < 16; y++)
			for (x = 0; x < 16; x++)
				sadm += abs(e->src_y[(y0 + y) * stride + x0 + x] - mean);
		if (sadm + 32 * lam < best_of_refine)
			e->cut_votes++;
	}
	/* the SAD of the chosen prediction = its cost without the vector's share */
	mad_qp(e, mx, my, (int)(best_of_refine - lam * (mv_bits_q(qx) + mv_bits_q(qy))));
	mb->type = H264O_MB_P16x16;
	mb->mvx = (int16_t)qx; /* quarter samples */
	mb->mvy = (int16_t)qy;
	mb->i16_mode = 0;
	mb->chroma_mode = 0;
	memset(mb->i4_modes, 2, sizeof(mb->i4_modes));
	memset(mb->nnz, 0, sizeof(mb->nnz));
	memset(mb->dc16, 0, sizeof(mb->dc16));
	mb->cbp = 0;

	for (idx = 0; idx < 16; idx++) {
		int bx = h264o_blk_x[idx], by = h264o_blk_y[idx], x, y, n;
		int16_t res[16];
		uint8_t *dst = ry + (y0 + 4 * by) * stride + x0 + 4 * bx;

		for (y = 0; y < 4; y++)
			for (x = 0; x < 4; x++) {
				int p = luma_sample_q(refy, stride, x0 + (qx >> 2) + 4 * bx + x, y0 + (qy >> 2) + 4 * by + y, qx & 3, qy & 3);

				res[4 * y + x] = (int16_t)(e->src_y[(y0 + 4 * by + y) * stride + x0 + 4 * bx + x] - p);
				dst[y * stride + x] = (uint8_t)p;
			}
		n = luma_block_16(e, res, mb->luma[idx], dst, stride, 0);
		mb->nnz[bx + 4 * by] = (uint8_t)n;
		if (n)
			mb->cbp |= (int8_t)(1 << (idx >> 2));
	}
	/* chroma motion compensation (8.4.2.2.2): mvC = luma mv in 1/8 chroma
	 * samples; whole-sample luma vectors give fractions 0 or 4 */
	{
		int mvcx = qx, mvcy = qy;
		int ix = mvcx >> 3, iy = mvcy >> 3, fx = mvcx & 7, fy = mvcy & 7;
		int x, y;

		for (c = 0; c < 2; c++)
			for (y = 0; y < 8; y++)
				for (x = 0; x < 8; x++) {
					int px = mx * 8 + x + ix, py = my * 8 + y + iy;
					/* the luma block is inside the picture, so px..px+1 and
					 * py..py+1 are inside whenever the fraction is non-zero */
					int px1 = fx ? px + 1 : px, py1 = fy ? py + 1 : py;
					int A = refuv[py * stride + 2 * px + c];
					int B = refuv[py * stride + 2 * px1 + c];
					int C = refuv[py1 * stride + 2 * px + c];
					int D = refuv[py1 * stride + 2 * px1 + c];

					predc[c][8 * y + x] = (uint8_t)(((8 - fx) * (8 - fy) * A + fx * (8 - fy) * B +
							(8 - fx) * fy * C + fx * fy * D + 32) >> 6);
				}
	}
	chroma_residual(e, mb, mx, my, predc, 0);

	/* I_PCM: here the levels are final (the prediction comes from the previous picture): the count holds for
	 * what CAVLC will write */
	if (pcm_allowed(e)) {
		int bits = PCM_HEADER_BITS;

		for (idx = 0; idx < 16; idx++)
			if (mb->cbp & (1 << (idx >> 2)))
				bits += pcm_block_bits(mb->luma[idx], 16, 0);
		if (mb->cbp & 0x30)
			for (c = 0; c < 2; c++)
				bits += pcm_block_bits(mb->cdc[c], 4, 1);
		if (mb->cbp & 0x20)
			for (c = 0; c < 2; c++) {
				int b;

				for (b = 0; b < 4; b++)
					bits += pcm_block_bits(mb->cac[c][b] + 1, 15, 0);
			}
		if (bits > PCM_LIMIT)
			pcm_make(e, mx, my);
	}
}

/* motion vector prediction for a 16x16 partition, all neighbours inter with
 * refIdx 0 (8.4.1.3). Availability: picture and slice (row band) bounds. */
static int median3(int a, int b, int c)
{
	int mn = a < b ? a : b, mx = a < b ? b : a;

	return c < mn ? mn : c > mx ? mx : c;
}

static int mb_is_intra(const h264o_mb *m)
{
	return m->type == H264O_MB_I4x4 || m->type == H264O_MB_I16x16 || m->type == H264O_MB_IPCM;
}

/* Cyclic intra refresh (cfg.intra_refresh = N): in the t-th P picture after an IDR the macroblocks with
 * (x + 2*y + t) % N == 0 are coded intra, so N consecutive P pictures renew every macroblock. With
 * N >= 3 neither the left nor the upper neighbour of a refresh macroblock is one itself: under
 * constrained_intra_pred both are "not available" and the prediction is the constant 128. */
static int refresh_mb(const h264o_enc *e, int mx, int my)
{
	return e->cfg.intra_refresh > 0 && e->frame_p_count > 0 &&
	       (mx + 2 * my + e->frame_p_count) % e->cfg.intra_refresh == 0;
}

/* An intra neighbour (refresh macroblocks) is available with refIdx -1 and a zero vector. */
static void mv_pred(const h264o_enc *e, int mx, int my, int row0, int *px, int *py, int *skx, int *sky)
{
	const h264o_mb *A = NULL, *B = NULL, *C = NULL;
	int ax = 0, ay = 0, bx = 0, by = 0, cx = 0, cy = 0, n;
	int ra, rb, rc;

	if (mx > 0)
		A = &e->mbs[my * e->mb_w + mx - 1];
	if (my > row0) {
		B = &e->mbs[(my - 1) * e->mb_w + mx];
		if (mx < e->mb_w - 1)
			C = &e->mbs[(my - 1) * e->mb_w + mx + 1];
		else if (mx > 0)
			C = &e->mbs[(my - 1) * e->mb_w + mx - 1]; /* D replaces C */
	}
	ra = A && !mb_is_intra(A);
	rb = B && !mb_is_intra(B);
	rc = C && !mb_is_intra(C);
	if (ra) { ax = A->mvx; ay = A->mvy; }
	if (rb) { bx = B->mvx; by = B->mvy; }
	if (rc) { cx = C->mvx; cy = C->mvy; }
	n = ra + rb + rc; /* neighbours with refIdx 0 */
	if (!B && !C && A) {
		*px = ax; *py = ay;
	} else if (n == 1) {
		*px = ra ? ax : rb ? bx : cx;
		*py = ra ? ay : rb ? by : cy;
	} else {
		*px = median3(ax, bx, cx);
		*py = median3(ay, by, cy);
	}
	/* P_Skip inference (8.4.1.1) */
	if (!A || !B || (ra && ax == 0 && ay == 0) || (rb && bx == 0 && by == 0)) {
		*skx = 0; *sky = 0;
	} else {
		*skx = *px; *sky = *py;
	}
}

/* ------------------------------------------------------------------ */
/* deblocking filter (clause 8.7), macroblock raster order, in place   */

/* boundary strength between 4x4 luma blocks (pbx,pby) of MB p and (qbx,qby) of MB q */
static int bs_of(const h264o_mb *p, int pbx, int pby, const h264o_mb *q, int qbx, int qby, int mb_edge)
{
	if (mb_is_intra(p) || mb_is_intra(q))
		return mb_edge ? 4 : 3;
	if (p->nnz[pbx + 4 * pby] || q->nnz[qbx + 4 * qby])
		return 2;
	if (abs(p->mvx - q->mvx) >= 4 || abs(p->mvy - q->mvy) >= 4) /* a whole sample or more (8.7.2.1) */
		return 1; /* full-pel vectors: any difference is >= 4 quarter samples */
	return 0;
}

/* filter one line of samples across an edge; pix points at q0, step = distance
 * between successive samples across the edge */
static void filter_luma_line(uint8_t *pix, int step, int bs, int alpha, int beta, int tc0)
{
	int p0 = pix[-step], p1 = pix[-2 * step], p2 = pix[-3 * step];
	int q0 = pix[0], q1 = pix[step], q2 = pix[2 * step];
	int ap, aq;

	if (abs(p0 - q0) >= alpha || abs(p1 - p0) >= beta || abs(q1 - q0) >= beta)
		return;
	ap = abs(p2 - p0);
	aq = abs(q2 - q0);
	if (bs < 4) {
		int tc = tc0 + (ap < beta) + (aq < beta);
		int delta = (((q0 - p0) * 4) + (p1 - q1) + 4) >> 3;

		delta = delta < -tc ? -tc : delta > tc ? tc : delta;
		pix[-step] = (uint8_t)clip255(p0 + delta);
		pix[0] = (uint8_t)clip255(q0 - delta);
		if (ap < beta) {
			int d = (p2 + ((p0 + q0 + 1) >> 1) - (p1 << 1)) >> 1;

			d = d < -tc0 ? -tc0 : d > tc0 ? tc0 : d;
			pix[-2 * step] = (uint8_t)(p1 + d);
		}
		if (aq < beta) {
			int d = (q2 + ((p0 + q0 + 1) >> 1) - (q1 << 1)) >> 1;

			d = d < -tc0 ? -tc0 : d > tc0 ? tc0 : d;
			pix[step] = (uint8_t)(q1 + d);
		}
	} else {
		int p3 = pix[-4 * step], q3 = pix[3 * step];
		int small = abs(p0 - q0) < ((alpha >> 2) + 2);

		if (ap < beta && small) {
			pix[-step] = (uint8_t)((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
			pix[-2 * step] = (uint8_t)((p2 + p1 + p0 + q0 + 2) >> 2);
			pix[-3 * step] = (uint8_t)((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
		} else {
			pix[-step] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
		}
		if (aq < beta && small) {
			pix[0] = (uint8_t)((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
			pix[step] = (uint8_t)((p0 + q0 + q1 + q2 + 2) >> 2);
			pix[2 * step] = (uint8_t)((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
		} else {
			pix[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
		}
	}
}

static void filter_chroma_line(uint8_t *pix, int step, int bs, int alpha, int beta, int tc0)
{
	int p0 = pix[-step], p1 = pix[-2 * step];
	int q0 = pix[0], q1 = pix[step];

	if (abs(p0 - q0) >= alpha || abs(p1 - p0) >= beta || abs(q1 - q0) >= beta)
		return;
	if (bs < 4) {
		int tc = tc0 + 1;
		int delta = (((q0 - p0) * 4) + (p1 - q1) + 4) >> 3;

		delta = delta < -tc ? -tc : delta > tc ? tc : delta;
		pix[-step] = (uint8_t)clip255(p0 + delta);
		pix[0] = (uint8_t)clip255(q0 - delta);
	} else {
		pix[-step] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
		pix[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
	}
}

static int chroma_qp_at(const h264o_enc *e, int qp_y)
{
	int q = qp_y + e->cfg.chroma_qp_offset;

	return h264o_chroma_qp[q < 0 ? 0 : q > 51 ? 51 : q];
}

static void deblock_mb(h264o_enc *e, int mx, int my, int row0)
{
	const h264o_mb *q = &e->mbs[my * e->mb_w + mx];
	uint8_t *ry = e->rec_y[e->cur], *ruv = e->rec_uv[e->cur];
	int stride = e->cw;
	/* thresholds come from the mean of the two macroblocks' QP_Y (and of their QP_C) across a macroblock
	 * edge (8.7.2.2); they only differ from picture to picture unless regions of interest are set */
	/* (8.7.2.2: qPp of an I_PCM macroblock is 0) */
	const int qq = q->type == H264O_MB_IPCM ? 0 : e->qp_eff[my * e->mb_w + mx], qqc = chroma_qp_at(e, qq);
	int qp = qq, qpc = qqc;
	int alpha_y = h264o_alpha[qp], beta_y = h264o_beta[qp];
	int alpha_c = h264o_alpha[qpc], beta_c = h264o_beta[qpc];
	int edge, k, c;

	/* vertical edges, left to right */
	for (edge = 0; edge < 4; edge++) {
		const h264o_mb *p = q;
		int bs[4];

		if (edge == 0) {
			if (mx == 0)
				continue;
			p = &e->mbs[my * e->mb_w + mx - 1];
		}
		{
			const int qn = edge == 0 ? (p->type == H264O_MB_IPCM ? 0 : e->qp_eff[my * e->mb_w + mx - 1]) : qq;

			qp = (qn + qq + 1) >> 1;
			qpc = (chroma_qp_at(e, qn) + qqc + 1) >> 1;
			alpha_y = h264o_alpha[qp]; beta_y = h264o_beta[qp];
			alpha_c = h264o_alpha[qpc]; beta_c = h264o_beta[qpc];
		}
		for (k = 0; k < 4; k++)
			bs[k] = edge == 0 ? bs_of(p, 3, k, q, 0, k, 1) : bs_of(q, edge - 1, k, q, edge, k, 0);
		for (k = 0; k < 16; k++) {
			int b = bs[k >> 2];

			if (b)
				filter_luma_line(ry + (my * 16 + k) * stride + mx * 16 + 4 * edge, 1, b,
						 alpha_y, beta_y, b < 4 ? h264o_tc0[qp][b - 1] : 0);
		}
		if (!(edge & 1)) /* chroma edges 0 and 2 <-> chroma columns 0 and 4 */
			for (c = 0; c < 2; c++)
				for (k = 0; k < 8; k++) {
					int b = bs[k >> 1];

					if (b)
						filter_chroma_line(ruv + (my * 8 + k) * stride + 2 * (mx * 8 + 2 * edge) + c,
								   2, b, alpha_c, beta_c, b < 4 ? h264o_tc0[qpc][b - 1] : 0);
				}
	}
	/* horizontal edges, top to bottom */
	for (edge = 0; edge < 4; edge++) {
		const h264o_mb *p = q;
		int bs[4];

		if (edge == 0) {
			if (my == row0) /* picture top, or slice top with idc 2 */
				continue;
			p = &e->mbs[(my - 1) * e->mb_w + mx];
		}
		{
			const int qn = edge == 0 ? (p->type == H264O_MB_IPCM ? 0 : e->qp_eff[(my - 1) * e->mb_w + mx]) : qq;

			qp = (qn + qq + 1) >> 1;
			qpc = (chroma_qp_at(e, qn) + qqc + 1) >> 1;
			alpha_y = h264o_alpha[qp]; beta_y = h264o_beta[qp];
			alpha_c = h264o_alpha[qpc]; beta_c = h264o_beta[qpc];
		}
		for (k = 0; k < 4; k++)
			bs[k] = edge == 0 ? bs_of(p, k, 3, q, k, 0, 1) : bs_of(q, k, edge - 1, q, k, edge, 0);
		for (k = 0; k < 16; k++) {
			int b = bs[k >> 2];

			if (b)
				filter_luma_line(ry + (my * 16 + 4 * edge) * stride + mx * 16 + k, stride, b,
						 alpha_y, beta_y, b < 4 ? h264o_tc0[qp][b - 1] : 0);
		}
		if (!(edge & 1))
			for (c = 0; c < 2; c++)
				for (k = 0; k < 8; k++) {
					int b = bs[k >> 1];

					if (b)
						filter_chroma_line(ruv + (my * 8 + 2 * edge) * stride + 2 * (mx * 8 + k) + c,
								   stride, b, alpha_c, beta_c, b < 4 ? h264o_tc0[qpc][b - 1] : 0);
				}
	}
}

/* ------------------------------------------------------------------ */
/* CAVLC (clause 9.2)                                                  */

static void cavlc_block(bitw *b, const int16_t *lv, int maxc, int nC)
{
	int pos[16], n = 0, t1 = 0, i, k, suffix_len, total_zeros, zeros_left;

	for (i = 0; i < maxc; i++)
		if (lv[i])
			pos[n++] = i;
	for (k = n - 1; k >= 0 && t1 < 3; k--) {
		if (lv[pos[k]] == 1 || lv[pos[k]] == -1)
			t1++;
		else
			break;
	}
	if (nC < 0)
		bw_put(b, h264o_chroma_dc_coeff_token_bits[4 * n + t1], h264o_chroma_dc_coeff_token_len[4 * n + t1]);
	else {
		int tab = nC < 2 ? 0 : nC < 4 ? 1 : nC < 8 ? 2 : 3;

		bw_put(b, h264o_coeff_token_bits[tab][4 * n + t1], h264o_coeff_token_len[tab][4 * n + t1]);
	}
	if (!n)
		return;
	for (k = n - 1; k > n - 1 - t1; k--)
		bw_put(b, lv[pos[k]] < 0, 1);
	suffix_len = (n > 10 && t1 < 3) ? 1 : 0;
	for (k = n - 1 - t1; k >= 0; k--) {
		int level = lv[pos[k]];
		int a = level < 0 ? -level : level;
		int code = 2 * a - 2 + (level < 0);

		if (k == n - 1 - t1 && t1 < 3)
			code -= 2;
		if (suffix_len == 0) {
			if (code < 14) {
				bw_put(b, 1, code + 1);
			} else if (code < 30) {
				bw_put(b, 1, 15);
				bw_put(b, (uint32_t)(code - 14), 4);
			} else {
				bw_put(b, 1, 16);
				bw_put(b, (uint32_t)(code - 30), 12);
			}
		} else {
			if (code < (15 << suffix_len)) {
				bw_put(b, 1, (code >> suffix_len) + 1);
				bw_put(b, (uint32_t)(code & ((1 << suffix_len) - 1)), suffix_len);
			} else {
				bw_put(b, 1, 16);
				bw_put(b, (uint32_t)(code - (15 << suffix_len)), 12);
			}
		}
		if (suffix_len == 0)
			suffix_len = 1;
		if (a > (3 << (suffix_len - 1)) && suffix_len < 6)
			suffix_len++;
	}
	total_zeros = pos[n - 1] + 1 - n;
	if (n < maxc) {
		if (maxc == 4)
			bw_put(b, h264o_chroma_dc_total_zeros_bits[n - 1][total_zeros],
			       h264o_chroma_dc_total_zeros_len[n - 1][total_zeros]);
		else
			bw_put(b, h264o_total_zeros_bits[n - 1][total_zeros],
			       h264o_total_zeros_len[n - 1][total_zeros]);
	}
	zeros_left = total_zeros;
	for (k = n - 1; k > 0 && zeros_left > 0; k--) {
		int run = pos[k] - pos[k - 1] - 1;
		int t = (zeros_left > 7 ? 7 : zeros_left) - 1;

		bw_put(b, h264o_run_before_bits[t][run], h264o_run_before_len[t][run]);
		zeros_left -= run;
	}
}

int h264o_cavlc_block(const int16_t *levels, int max_coeff, int nC, uint8_t *out, int cap)
{
	bitw b;

	bw_init(&b, out, cap);
	cavlc_block(&b, levels, max_coeff, nC);
	return b.overflow ? -ENOSPC : (int)b.pos;
}

/* nC for a block from the total_coeff of its left (A) and top (B) neighbours
 * (9.2.1). plane: 0 luma (4x4 grid), 1 Cb, 2 Cr (2x2 grid). */
static int calc_nc(const h264o_enc *e, int mx, int my, int row0, int plane, int bx, int by)
{
	int w = plane ? 2 : 4, base = plane ? 16 + 4 * (plane - 1) : 0;
	const h264o_mb *cur = &e->mbs[my * e->mb_w + mx];
	int na = -1, nb = -1;

	if (bx > 0)
		na = cur->nnz[base + (bx - 1) + w * by];
	else if (mx > 0)
		na = e->mbs[my * e->mb_w + mx - 1].nnz[base + (w - 1) + w * by];
	if (by > 0)
		nb = cur->nnz[base + bx + w * (by - 1)];
	else if (my > row0)
		nb = e->mbs[(my - 1) * e->mb_w + mx].nnz[base + bx + w * (w - 1)];
	if (na >= 0 && nb >= 0)
		return (na + nb + 1) >> 1;
	if (na >= 0)
		return na;
	if (nb >= 0)
		return nb;
	return 0;
}

static void write_residual(const h264o_enc *e, bitw *b, const h264o_mb *mb, int mx, int my, int row0)
{
	int i8, i4, c, blk;

	if (mb->type == H264O_MB_I16x16)
		cavlc_block(b, mb->dc16, 16, calc_nc(e, mx, my, row0, 0, 0, 0));
	for (i8 = 0; i8 < 4; i8++) {
		if (!(mb->cbp & (1 << i8)))
			continue;
		for (i4 = 0; i4 < 4; i4++) {
			int idx = 4 * i8 + i4;
			int nC = calc_nc(e, mx, my, row0, 0, h264o_blk_x[idx], h264o_blk_y[idx]);

			if (mb->type == H264O_MB_I16x16)
				cavlc_block(b, mb->luma[idx] + 1, 15, nC);
			else
				cavlc_block(b, mb->luma[idx], 16, nC);
		}
	}
	if (mb->cbp & 0x30) {
		for (c = 0; c < 2; c++)
			cavlc_block(b, mb->cdc[c], 4, -1);
		if (mb->cbp & 0x20)
			for (c = 0; c < 2; c++)
				for (blk = 0; blk < 4; blk++)
					cavlc_block(b, mb->cac[c][blk] + 1, 15,
						    calc_nc(e, mx, my, row0, 1 + c, blk & 1, blk >> 1));
	}
}

/* predicted Intra4x4PredMode for block idx (8.3.1.1) */
static int pred_i4_mode(const h264o_enc *e, int mx, int my, int row0, int idx)
{
	int bx = h264o_blk_x[idx], by = h264o_blk_y[idx];
	const h264o_mb *cur = &e->mbs[my * e->mb_w + mx];
	const h264o_mb *ma = cur, *mb_ = cur;
	int a, b, k;
	int abx = bx - 1, aby = by, bbx = bx, bby = by - 1;

	if (bx == 0) {
		if (mx == 0)
			return 2;
		ma = &e->mbs[my * e->mb_w + mx - 1];
		abx = 3;
	}
	if (by == 0) {
		if (my == row0)
			return 2;
		mb_ = &e->mbs[(my - 1) * e->mb_w + mx];
		bby = 3;
	}
	a = b = 2;
	if (ma->type == H264O_MB_I4x4)
		for (k = 0; k < 16; k++)
			if (h264o_blk_x[k] == abx && h264o_blk_y[k] == aby)
				a = ma->i4_modes[k];
	if (mb_->type == H264O_MB_I4x4)
		for (k = 0; k < 16; k++)
			if (h264o_blk_x[k] == bbx && h264o_blk_y[k] == bby)
				b = mb_->i4_modes[k];
	return a < b ? a : b;
}

/* mb_qp_delta (7.4.5): against QP_Y of the macroblock before it in decoding order, the slice's at its start */
static int qp_delta_of(const h264o_enc *e, int mx, int my, int row0)
{
	const int i = my * e->mb_w + mx;
	int d = e->qp_t[i] - (i == row0 * e->mb_w ? e->qp : e->qp_eff[i - 1]);

	/* 7.4.5: mb_qp_delta lies in -26..+25 and QP_Y = (QP_Y,pred + mb_qp_delta + 52) % 52: a step of
	 * more than that (two regions at +24 and -24 side by side) goes round the other way */
	if (d > 25)
		d -= 52;
	else if (d < -26)
		d += 52;
	return d;
}

static void write_intra_mb(const h264o_enc *e, bitw *b, const h264o_mb *mb, int mx, int my,
			   int row0, int p_slice)
{
	int idx;

	if (mb->type == H264O_MB_IPCM) { /* 7.3.5: mb_type 25, pcm_alignment_zero_bits, 256 + 2 x 64 samples */
		int x, y, c;

		bw_ue(b, (uint32_t)(25 + (p_slice ? 5 : 0)));
		while (b->pos & 7)
			bw_put(b, 0, 1);
		for (y = 0; y < 16; y++)
			for (x = 0; x < 16; x++)
				bw_put(b, e->src_y[(size_t)(my * 16 + y) * e->cw + mx * 16 + x], 8);
		for (c = 0; c < 2; c++)
			for (y = 0; y < 8; y++)
				for (x = 0; x < 8; x++)
					bw_put(b, e->src_uv[(size_t)(my * 8 + y) * e->cw + 2 * (mx * 8 + x) + c], 8);
		return;
	}
	if (mb->type == H264O_MB_I4x4) {
		bw_ue(b, (uint32_t)(p_slice ? 5 : 0));
		for (idx = 0; idx < 16; idx++) {
			int pm = pred_i4_mode(e, mx, my, row0, idx), m = mb->i4_modes[idx];

			if (m == pm)
				bw_put(b, 1, 1);
			else {
				bw_put(b, 0, 1);
				bw_put(b, (uint32_t)(m < pm ? m : m - 1), 3);
			}
		}
		bw_ue(b, (uint32_t)mb->chroma_mode);
		bw_ue(b, e->cbp_code_intra[mb->cbp & 0x3f]);
		if (mb->cbp & 0x3f)
			bw_se(b, qp_delta_of(e, mx, my, row0)); /* mb_qp_delta */
	} else {
		int t = 1 + mb->i16_mode + 4 * ((mb->cbp >> 4) & 3) + ((mb->cbp & 15) ? 12 : 0);

		bw_ue(b, (uint32_t)(t + (p_slice ? 5 : 0)));
		bw_ue(b, (uint32_t)mb->chroma_mode);
		bw_se(b, qp_delta_of(e, mx, my, row0)); /* mb_qp_delta */
	}
	write_residual(e, b, mb, mx, my, row0);
}

/* ------------------------------------------------------------------ */
/* session                                                             */

void h264o_default_config(h264o_config *c, int width, int height)
{
	memset(c, 0, sizeof(*c));
	c->width = width;
	c->height = height;
	c->qp = 24;            /* userspace/h264enc.c:63 */
	c->keyint = 25;        /* userspace/h264enc.c:64 */
	c->profile_idc = 66;   /* north_star: baseline; reference asks 77 (h264enc.c:61) */
	c->level_idc = (ALIGN16(width) / 16) * (ALIGN16(height) / 16) > 8192 ? 51 : 41;
	c->search_range = 16;
	c->slice_count = 1;
	c->chroma_qp_offset = 4; /* cedar.c:971-972 */
}

h264o_enc *h264o_open(const h264o_config *cfg, int *err)
{
	h264o_enc *e;
	int i, dummy;

	if (!err)
		err = &dummy;
	*err = -EINVAL;
	if (!cfg)
		return NULL;
	/* cedar.c:749-789 validation, with dst = ALIGN16(src) implied */
	if (cfg->width <= 0 || cfg->height <= 0 || (cfg->width & 1) || (cfg->height & 1))
		return NULL;
	if (cfg->qp <= 0 || cfg->qp > 47)
		return NULL;
	if (cfg->keyint <= 0 || cfg->keyint >= 32)
		return NULL;
	if (cfg->subpel < 0 || cfg->subpel > 2)
		return NULL;
	if (cfg->roi_count < 0 || cfg->roi_count > 4)
		return NULL;
	for (i = 0; i < cfg->roi_count; i++)
		if (cfg->roi[i][0] < 0 || cfg->roi[i][1] < 0 || cfg->roi[i][2] <= 0 || cfg->roi[i][3] <= 0 ||
		    cfg->roi[i][4] < -24 || cfg->roi[i][4] > 24)
			return NULL;
	for (i = 0; i < 5; i++)
		if (cfg->mad_dqp[i] < -12 || cfg->mad_dqp[i] > 12)
			return NULL;
	for (i = 0; i < 4; i++)
		if (cfg->mad_th[i] < 0 || cfg->mad_th[i] > 255 || (i && cfg->mad_th[i] < cfg->mad_th[i - 1]))
			return NULL;
	if (cfg->intra_refresh < 0 || (cfg->intra_refresh > 0 && cfg->intra_refresh < 3) || cfg->intra_refresh > 65535)
		return NULL;
	if (cfg->scene_cut < 0 || cfg->scene_cut > 100 || cfg->pcm < -1 || cfg->pcm > 0 || cfg->rc_lanes < 0 ||
	    (cfg->rc_lanes > 1 && (cfg->scene_cut > 0 || cfg->strict_mb_bits == 2)))
		return NULL;
	if (cfg->search_range < 0 || cfg->search_range > 16)
		return NULL;
	if (cfg->slice_count < 1 || cfg->slice_count > ALIGN16(cfg->height) / 16)
		return NULL;
	if (cfg->chroma_qp_offset < -12 || cfg->chroma_qp_offset > 12) /* H.264 7.4.2.2 */
		return NULL;
	if (cfg->fps_num < 0 || cfg->fps_den < 0 || (cfg->fps_num > 0) != (cfg->fps_den > 0) ||
	    cfg->fps_num > 0x3fffffff)
		return NULL;
	*err = -ENOMEM;
	e = calloc(1, sizeof(*e));
	if (!e)
		return NULL;
	e->cfg = *cfg;
	for (i = 0; i < 5; i++)
		if (cfg->mad_dqp[i])
			e->mad_on = 1;
	e->cw = ALIGN16(cfg->width);
	e->ch = ALIGN16(cfg->height);
	e->mb_w = e->cw / 16;
	e->mb_h = e->ch / 16;
	e->closed_loop = getenv("H264O_CLOSED_LOOP_INTRA") != NULL && atoi(getenv("H264O_CLOSED_LOOP_INTRA")) != 0;
	e->src_y = malloc((size_t)e->cw * e->ch);
	e->src_uv = malloc((size_t)e->cw * e->ch / 2);
	for (i = 0; i < 2; i++) {
		e->rec_y[i] = calloc((size_t)e->cw * e->ch, 1);
		e->rec_uv[i] = calloc((size_t)e->cw * e->ch / 2, 1);
	}
	e->mbs = calloc((size_t)e->mb_w * e->mb_h, sizeof(h264o_mb));
	e->qp_t = calloc((size_t)e->mb_w * e->mb_h, 1);
	e->qp_eff = calloc((size_t)e->mb_w * e->mb_h, 1);
	e->rbsp_cap = e->mb_w * e->mb_h * 2048 + 1024;
	e->rbsp = malloc((size_t)e->rbsp_cap);
	if (!e->src_y || !e->src_uv || !e->rec_y[0] || !e->rec_y[1] || !e->rec_uv[0] ||
	    !e->rec_uv[1] || !e->mbs || !e->rbsp || !e->qp_t || !e->qp_eff) {
		h264o_close(e);
		return NULL;
	}
	for (i = 0; i < 48; i++) {
		e->cbp_code_intra[h264o_cbp_intra_from_code[i]] = (uint8_t)i;
		e->cbp_code_inter[h264o_cbp_inter_from_code[i]] = (uint8_t)i;
	}
	*err = 0;
	return e;
}

void h264o_close(h264o_enc *e)
{
	int i;

	if (!e)
		return;
	free(e->src_y);
	free(e->src_uv);
	for (i = 0; i < 2; i++) {
		free(e->rec_y[i]);
		free(e->rec_uv[i]);
	}
	free(e->mbs);
	free(e->qp_t);
	free(e->qp_eff);
	free(e->rbsp);
	free(e);
}

int h264o_max_frame_bytes(const h264o_enc *e)
{
	return e->mb_w * e->mb_h * 3072 + 4096;
}

int h264o_mb_width(const h264o_enc *e) { return e->mb_w; }
int h264o_mb_height(const h264o_enc *e) { return e->mb_h; }
const h264o_mb *h264o_mbs(const h264o_enc *e) { return e->mbs; }
/* after the post-encode swap (cedar.c:1198-1201) the last picture is "previous" */
const uint8_t *h264o_recon_luma(const h264o_enc *e) { return e->rec_y[e->cur ^ 1]; }
const uint8_t *h264o_recon_chroma(const h264o_enc *e) { return e->rec_uv[e->cur ^ 1]; }
int h264o_last_frame_type(const h264o_enc *e) { return e->last_idr; }
int h264o_max_mb_bits(const h264o_enc *e) { return e->max_mb_bits; }
int h264o_frame_qp(const h264o_enc *e) { return e->qp; }
const uint8_t *h264o_mb_qp(const h264o_enc *e) { return e->qp_eff; }

/* ---- frame-level rate control (the VE has RC registers the driver never programs, cedar_regs.h:43-53;
 * SURVEY 8(f)-4). One QP per picture, carried by slice_qp_delta. The law, integer only:
 *   - frame n may use the sizes of frames <= n - 1 - 16 only (a session keeps up to 16 frames in
 *     flight: whatever the caller's pipelining, those frames have been collected), over a window of
 *     the last keyint of them;
 *   - every known frame is scaled to what it would have cost at the configured QP, by the model
 *     "6 QP steps halve the bits": bits * 2^((qp_i - qp0) / 6), in Q16;
 *   - the new QP is qp0 + d with d the smallest step for which 2^(d/6) * target >= that average:
 *     the lowest QP at which the model predicts no more than the target, clamped to 1..47 (cedar.c:771).
 * 2^(d/6) in Q16 for d = 6q + r: */
static int64_t rc_pow2_6_q16(int d)
{
	static const int64_t t[6] = { 65536, 73562, 82570, 92682, 104032, 116772 };
	int q = d >= 0 ? d / 6 : -((-d + 5) / 6), r = d - 6 * q;

	return q >= 0 ? t[r] << q : t[r] >> -q;
}

#define RC_LAG 16
static int rc_pick_qp(int qp0, int64_t target, const int64_t *bits, const int *qps, long n, int keyint)
{
	long known = n - RC_LAG, w, i;
	int64_t sum = 0, avg;
	int d = -46, qp;

	if (known <= 0 || target <= 0)
		return qp0;
	w = known < keyint ? known : keyint;
	for (i = known - w; i < known; i++)
		sum += (bits[i & 63] * rc_pow2_6_q16(qps[i & 63] - qp0)) >> 16;
	avg = sum / w;
	while (d < 46 && rc_pow2_6_q16(d) * target < (avg << 16))
		d++;
	qp = qp0 + d;
	return qp < 1 ? 1 : qp > 47 ? 47 : qp;
}

/* The law of a session that codes G closed GOPs side by side (h264o_config.rc_lanes; gop_parallel of the HIP path, whose
 * lock-step step q holds frame q % keyint of the G GOPs of group q / keyint): frames of later GOPs of a group are coded
 * beside the earlier ones, so nothing inside a group can steer it. ONE QP per group, from the mean bits per frame (scaled
 * to the configured QP as above) of the group RC_GROUP_LAG groups back: the nearest group all of whose steps have been
 * collected when the first step of this one is submitted, whatever the caller's pipelining (16 steps in flight at most). */
static int rc_group_lag(int keyint)
{
	return 1 + (RC_LAG + keyint - 1) / keyint;
}

static int rc_pick_qp_group(int qp0, int64_t target, const int64_t *sum, const int *cnt, const long *id1, long m, int keyint)
{
	const long r = m - rc_group_lag(keyint);
	int64_t avg;
	int d = -46, qp;

	if (r < 0 || target <= 0 || id1[r & 31] != r + 1 || cnt[r & 31] <= 0)
		return qp0;
	avg = sum[r & 31] / cnt[r & 31];
	while (d < 46 && rc_pow2_6_q16(d) * target < (avg << 16))
		d++;
	qp = qp0 + d;
	return qp < 1 ? 1 : qp > 47 ? 47 : qp;
}

static void pad_source(h264o_enc *e, const uint8_t *luma, const uint8_t *chroma)
{
	int w = e->cfg.width, h = e->cfg.height, x, y;

	for (y = 0; y < e->ch; y++) {
		int sy = y < h ? y : h - 1;

		memcpy(e->src_y + (size_t)y * e->cw, luma + (size_t)sy * w, (size_t)w);
		for (x = w; x < e->cw; x++)
			e->src_y[(size_t)y * e->cw + x] = luma[(size_t)sy * w + w - 1];
	}
	for (y = 0; y < e->ch / 2; y++) {
		int sy = y < h / 2 ? y : h / 2 - 1;

		memcpy(e->src_uv + (size_t)y * e->cw, chroma + (size_t)sy * w, (size_t)w);
		for (x = w; x < e->cw; x += 2) {
			e->src_uv[(size_t)y * e->cw + x] = chroma[(size_t)sy * w + w - 2];
			e->src_uv[(size_t)y * e->cw + x + 1] = chroma[(size_t)sy * w + w - 1];
		}
	}
}

int h264o_encode(h264o_enc *e, const uint8_t *luma, const uint8_t *chroma, uint8_t *out, int out_cap)
{
	int max_mb_bits = 0, ue_bits_of_run = 0;
	int idr, forced_idr = 0;
	int o, o_hdr, s, mx, my, r;
	int idr_pic_id;

	pad_source(e, luma, chroma);
restart: /* cfg.scene_cut: the analysis of a P picture may ask for an IDR picture instead */
	if (forced_idr)
		e->frame_p_count = 0; /* a new GOP starts here */
	idr = e->frame_p_count == 0; /* cedar.c:1047-1050 */
	/* (two IDR pictures in a row must differ in idr_pic_id: by the GOP's number on the regular cadence, by the
	 * count of IDR pictures where cuts put them anywhere) */
	idr_pic_id = (e->cfg.scene_cut > 0 ? e->idr_count : e->frame_count / e->cfg.keyint) & 1;
	o = 0;

	if (e->cfg.aud) {
		r = h264o_write_aud(idr, out + o, out_cap - o);
		if (r < 0)
			return r;
		o += r;
	}
	if (e->frame_count == 0 || (idr && e->cfg.repeat_headers)) { /* cedar.c:1058-1061 */
		r = h264o_write_sps(&e->cfg, out + o, out_cap - o);
		if (r < 0)
			return r;
		o += r;
		r = h264o_write_pps(&e->cfg, out + o, out_cap - o);
		if (r < 0)
			return r;
		o += r;
	}

	e->qp = e->cfg.qp;
	if (e->cfg.bitrate > 0) {
		const int64_t fn = e->cfg.fps_num > 0 && e->cfg.fps_den > 0 ? e->cfg.fps_num : 30;
		const int64_t fd = e->cfg.fps_num > 0 && e->cfg.fps_den > 0 ? e->cfg.fps_den : 1;

		if (e->cfg.rc_lanes > 1)
			e->qp = rc_pick_qp_group(e->cfg.qp, (int64_t)e->cfg.bitrate * fd / fn, e->grp_sum, e->grp_n, e->grp_id1,
						 e->frame_count / ((long)e->cfg.rc_lanes * e->cfg.keyint), e->cfg.keyint);
		else
			e->qp = rc_pick_qp(e->cfg.qp, (int64_t)e->cfg.bitrate * fd / fn, e->rc_bits, e->rc_qp, e->frame_count, e->cfg.keyint);
	}
	o_hdr = o;
again: /* strict_mb_bits == 2: a picture with a macroblock above 3200 bits is coded again, 6 QP steps coarser */
	o = o_hdr;
	max_mb_bits = 0;
	e->cut_votes = 0;
	/* regions of interest: the first one that covers a macroblock moves its QP */
	for (my = 0; my < e->mb_h; my++)
		for (mx = 0; mx < e->mb_w; mx++) {
			int q = e->qp, k;

			for (k = 0; k < e->cfg.roi_count; k++) {
				const int *r = e->cfg.roi[k];

				if (mx >= r[0] && mx < r[0] + r[2] && my >= r[1] && my < r[1] + r[3]) {
					q += r[4];
					break;
				}
			}
			e->qp_t[my * e->mb_w + mx] = (uint8_t)(q < 1 ? 1 : q > 51 ? 51 : q);
		}
	e->mbqp = e->qp;

	/* analysis + reconstruction */
	for (s = 0; s < e->cfg.slice_count; s++) {
		int r0, r1;

		slice_rows(e, s, &r0, &r1);
		if (idr && e->closed_loop) {
			/* NOT the encoder policy (the decision is open loop so that it is
			 * macroblock-parallel): decide every macroblock on its reconstructed neighbours, in order */
			for (my = r0; my < r1; my++)
				for (mx = 0; mx < e->mb_w; mx++) {
					int y;

					for (y = 0; y < 16; y++)
						memcpy(e->rec_y[e->cur] + (size_t)(my * 16 + y) * e->cw + mx * 16,
						       e->src_y + (size_t)(my * 16 + y) * e->cw + mx * 16, 16);
					for (y = 0; y < 8; y++)
						memcpy(e->rec_uv[e->cur] + (size_t)(my * 8 + y) * e->cw + mx * 16,
						       e->src_uv + (size_t)(my * 8 + y) * e->cw + mx * 16, 16);
					e->mbqp = e->qp_t[my * e->mb_w + mx];
					intra_decide_mb(e, mx, my, r0);
					intra_recon_mb(e, mx, my, r0, 0);
					force_row_start_coded(e, &e->mbs[my * e->mb_w + mx], mx);
				}
		} else if (idr) {
			for (my = r0; my < r1; my++)
				for (mx = 0; mx < e->mb_w; mx++) {
					e->mbqp = e->qp_t[my * e->mb_w + mx];
					intra_decide_mb(e, mx, my, r0);
				}
			for (my = r0; my < r1; my++)
				for (mx = 0; mx < e->mb_w; mx++) {
					e->mbqp = e->qp_t[my * e->mb_w + mx];
					intra_recon_mb(e, mx, my, r0, 0);
					force_row_start_coded(e, &e->mbs[my * e->mb_w + mx], mx);
				}
		} else {
			for (my = r0; my < r1; my++)
				for (mx = 0; mx < e->mb_w; mx++) {
					h264o_mb *mb = &e->mbs[my * e->mb_w + mx];

					e->mbqp = e->qp_t[my * e->mb_w + mx];
					if (!refresh_mb(e, mx, my)) {
						inter_mb(e, mx, my, r0, r1);
						force_row_start_coded(e, mb, mx);
						continue;
					}
					{
						int sad128 = 0, x, y;

						for (y = 0; y < 16; y++)
							for (x = 0; x < 16; x++)
								sad128 += abs(e->src_y[(my * 16 + y) * e->cw + mx * 16 + x] - 128);
						mad_qp(e, mx, my, sad128);
					}
					mb->type = H264O_MB_I16x16;
					mb->i16_mode = 2;    /* DC: the one mode that needs no neighbour */
					mb->chroma_mode = 0; /* DC */
					mb->mvx = mb->mvy = 0;
					memset(mb->i4_modes, 2, sizeof(mb->i4_modes));
					intra_recon_mb(e, mx, my, r0, 1);
				}
			/* skip decision needs the whole vector field */
			for (my = r0; my < r1; my++)
				for (mx = 0; mx < e->mb_w; mx++) {
					h264o_mb *mb = &e->mbs[my * e->mb_w + mx];
					int px, py, skx, sky;

					if (mb_is_intra(mb))
						continue;
					mv_pred(e, mx, my, r0, &px, &py, &skx, &sky);
					if (mb->cbp == 0 && mb->mvx == skx && mb->mvy == sky)
						mb->type = H264O_MB_PSKIP;
				}
		}
	}

	if (!idr && e->cfg.scene_cut > 0 && e->cut_votes * 100 >= e->cfg.scene_cut * e->mb_w * e->mb_h) {
		forced_idr = 1;
		goto restart;
	}

	/* QP_Y as the decoder will hold it: only a macroblock with an mb_qp_delta (Intra16x16, or any coded
	 * block) can change it; the others -- P_Skip and I_PCM included: no mb_qp_delta means 0 -- inherit their
	 * predecessor's (7.4.5). (Only the deblocking filter takes an I_PCM macroblock's QP as 0, 8.7.2.2.) */
	for (s = 0; s < e->cfg.slice_count; s++) {
		int r0, r1, prev = e->qp;

		slice_rows(e, s, &r0, &r1);
		for (my = r0; my < r1; my++)
			for (mx = 0; mx < e->mb_w; mx++) {
				const h264o_mb *mb = &e->mbs[my * e->mb_w + mx];

				if (mb->type == H264O_MB_I16x16 || (mb->type != H264O_MB_IPCM && (mb->cbp & 0x3f)))
					prev = e->qp_t[my * e->mb_w + mx];
				e->qp_eff[my * e->mb_w + mx] = (uint8_t)prev;
			}
	}

	/* in-loop deblocking of the whole picture in macroblock order */
	for (my = 0; my < e->mb_h; my++) {
		int r0, r1;

		slice_rows(e, slice_of_row(e, my), &r0, &r1);
		for (mx = 0; mx < e->mb_w; mx++)
			deblock_mb(e, mx, my, r0);
	}

	/* entropy coding, one NAL per slice */
	for (s = 0; s < e->cfg.slice_count; s++) {
		int r0, r1, skip_run = 0;
		bitw b;

		slice_rows(e, s, &r0, &r1);
		bw_init(&b, e->rbsp, e->rbsp_cap);
		slice_header_bits_qp(&e->cfg, &b, idr, e->frame_p_count, idr_pic_id, r0 * e->mb_w, e->qp - e->cfg.qp);
		for (my = r0; my < r1; my++)
			for (mx = 0; mx < e->mb_w; mx++) {
				const h264o_mb *mb = &e->mbs[my * e->mb_w + mx];
				const uint64_t mb_start = b.pos;

				if (idr) {
					write_intra_mb(e, &b, mb, mx, my, r0, 0);
					if ((int)(b.pos - mb_start) > max_mb_bits)
						max_mb_bits = (int)(b.pos - mb_start);
					continue;
				}
				if (mb->type == H264O_MB_PSKIP) {
					skip_run++;
					continue;
				}
				bw_ue(&b, (uint32_t)skip_run);
				ue_bits_of_run = (int)(b.pos - mb_start);
				skip_run = 0;
				if (mb_is_intra(mb)) {
					write_intra_mb(e, &b, mb, mx, my, r0, 1); /* mb_type 5 + the I slice's */
				} else {
					int px, py, skx, sky;

					mv_pred(e, mx, my, r0, &px, &py, &skx, &sky);
					bw_ue(&b, 0); /* mb_type P_L0_16x16 */
					bw_se(&b, mb->mvx - px); /* vectors are kept in quarter samples */
					bw_se(&b, mb->mvy - py);
					bw_ue(&b, e->cbp_code_inter[mb->cbp & 0x3f]);
					if (mb->cbp & 0x3f)
						bw_se(&b, qp_delta_of(e, mx, my, r0)); /* mb_qp_delta */
					write_residual(e, &b, mb, mx, my, r0);
				}
				/* macroblock_layer() only: mb_skip_run belongs to slice_data() */
				if ((int)(b.pos - mb_start) - ue_bits_of_run > max_mb_bits)
					max_mb_bits = (int)(b.pos - mb_start) - ue_bits_of_run;
			}
		if (!idr && skip_run)
			bw_ue(&b, (uint32_t)skip_run);
		bw_trailing(&b);
		if (b.overflow)
			return -ENOSPC;
		/* H.264 A.3.1: no macroblock_layer() above 3200 bits. Sessions that code such macroblocks I_PCM (cfg.pcm,
		 * DESIGN.md 3.6) never get here; the others requantise the picture (2), refuse it (1: state untouched, as for
		 * any failed frame) or emit it */
		if (max_mb_bits > 3200 && e->cfg.strict_mb_bits == 2 && e->qp + 6 <= 51) {
			e->qp += 6; /* nothing of the session's state has been committed yet: just code it again */
			goto again;
		}
		if (max_mb_bits > 3200 && e->cfg.strict_mb_bits)
			return -EOVERFLOW;
		r = nal_write(out + o, out_cap - o, idr ? 3 : 2, idr ? 5 : 1, e->rbsp, (int)(b.pos >> 3));
		if (r < 0)
			return r;
		o += r;
	}

	e->rc_bits[e->frame_count & 63] = (int64_t)o * 8;
	e->rc_qp[e->frame_count & 63] = e->qp;
	if (e->cfg.bitrate > 0 && e->cfg.rc_lanes > 1) {
		const long m = e->frame_count / ((long)e->cfg.rc_lanes * e->cfg.keyint);

		if (e->grp_id1[m & 31] != m + 1) {
			e->grp_id1[m & 31] = m + 1;
			e->grp_sum[m & 31] = 0;
			e->grp_n[m & 31] = 0;
		}
		e->grp_sum[m & 31] += ((int64_t)o * 8 * rc_pow2_6_q16(e->qp - e->cfg.qp)) >> 16;
		e->grp_n[m & 31]++;
	}
	/* cedar.c:1193-1201 */
	e->max_mb_bits = max_mb_bits;
	e->last_idr = idr;
	if (idr)
		e->idr_count++;
	e->frame_p_count++;
	if (e->frame_p_count == e->cfg.keyint)
		e->frame_p_count = 0;
	e->frame_count++;
	e->cur ^= 1;
	return o;
}

/* ------------------------------------------------------------------ */
/* synthetic inputs                                   */

void h264o_gen_frame(int pattern, int w, int h, int frame, unsigned seed_offset,
		     uint8_t *luma, uint8_t *chroma)
{
	int x, y;

	if (pattern == 'A') { /* thumb.c:139-169 idea: static picture */
		for (y = 0; y < h; y++)
			for (x = 0; x < w; x++)
				luma[y * w + x] = (x < 4 || y < 4 || x >= w - 4 || y >= h - 4) ? 0xEB : 0x10;
		for (y = 0; y < h / 2; y++)
			for (x = 0; x < w / 2; x++) {
				chroma[y * w + 2 * x] = (uint8_t)(x & 0xFF);
				chroma[y * w + 2 * x + 1] = (uint8_t)(y & 0xFF);
			}
	} else if (pattern == 'B') { /* moving texture, global motion (3,2) */
		uint32_t s = 0x5EED0000u + (uint32_t)frame + seed_offset;

		for (y = 0; y < h; y++)
			for (x = 0; x < w; x++) {
				int u = x + 3 * frame, v = y + 2 * frame;

				s = s * 1664525u + 1013904223u;
				luma[y * w + x] = (uint8_t)((5 * u + 3 * v + 17 * ((u >> 4) ^ (v >> 4)) + (int)(s >> 29)) & 0xFF);
			}
		for (y = 0; y < h / 2; y++)
			for (x = 0; x < w / 2; x++) {
				int u = 2 * x + 3 * frame, v = 2 * y + 2 * frame;

				chroma[y * w + 2 * x] = (uint8_t)((128 + ((u >> 1) & 63) - 32) & 0xFF);
				chroma[y * w + 2 * x + 1] = (uint8_t)((128 + ((v >> 1) & 63) - 32) & 0xFF);
			}
	} else { /* 'C': noise */
		uint32_t s = 0xC0FFEE00u + (uint32_t)frame + seed_offset;
		int n = w * h;

		for (x = 0; x < n; x++) {
			s = s * 1664525u + 1013904223u;
			luma[x] = (uint8_t)(s >> 24);
		}
		for (x = 0; x < n / 2; x++) {
			s = s * 1664525u + 1013904223u;
			chroma[x] = (uint8_t)(s >> 24);
		}
	}
}
