/*
 * oracle/h264o_dec.c -- minimal H.264 baseline/CAVLC decoder, TEST ONLY.
 *
 * Purpose: the container has no H.264 decoder of any kind (SURVEY.md 0.3,
 * 8(c)), so the only conformance check available is a decoder written here:
 * tests decode the emitted Annex-B stream and require the decoded pictures to
 * equal the encoder's in-loop reconstruction sample for sample.
 *
 * It is written independently of h264o_enc.c (its own prediction, inverse
 * transform, motion compensation and deblocking code, in the spec's p[x,y]
 * formula style) and shares only the constant tables. It supports exactly
 * the syntax a baseline encoder of this kind can emit: frame pictures, CAVLC,
 * one reference, I (I4x4 / I16x16) and P (P_L0_16x16 / P_Skip / intra) MBs,
 * full-pel luma vectors, several slices, deblocking idc 0/1/2.
 * Anything else is reported as an error, never silently accepted.
 *
 * PARITY UNPINNED: the reference (libv/cedarx_h264_encoder) contains no decoder
 * and no test vectors; this follows ITU-T H.264 clauses 7.3, 8.3-8.7, 9.2.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "h264o.h"
#include "h264o_tables.h"

typedef struct dmb {
	int8_t type;      /* H264O_MB_* */
	int8_t slice;     /* slice number within the picture */
	int8_t qp;
	int8_t pad;
	int8_t i4[16];    /* by raster 4x4 position bx+4*by */
	uint8_t nnz[24];  /* luma raster 16, Cb 4, Cr 4 */
	int16_t mvx, mvy; /* quarter-pel */
} dmb;

struct h264o_dec {
	int have_sps, have_pps;
	int mb_w, mb_h, cw, ch, crop_r, crop_b;
	uint32_t vui_ticks, vui_scale; /* 0: no timing in the SPS */
	int vui_reorder, n_aud;
	int log2_max_frame_num, poc_type;
	int init_qp, chroma_qp_offset, deblock_ctrl, constrained_intra;
	uint8_t *cur_y, *cur_uv, *ref_y, *ref_uv;
	dmb *mbs;
	int *slice_idc, *slice_a, *slice_b; /* per slice deblock params */
	int n_slices, mbs_done, have_ref;
	uint8_t *rbsp;
	int rbsp_cap;
	char err[160];
};

/* ---------------- bit reader ---------------- */
typedef struct br {
	const uint8_t *p;
	int nbits; /* position of the rbsp stop bit = number of payload bits */
	int pos;
	int err;
} br;

static unsigned br_bit(br *b)
{
	unsigned v;

	if (b->pos >= b->nbits + 1) { /* allow reading the stop bit itself */
		b->err = 1;
		return 0;
	}
	v = (b->p[b->pos >> 3] >> (7 - (b->pos & 7))) & 1;
	b->pos++;
	return v;
}

static unsigned br_u(br *b, int n)
{
	unsigned v = 0;

	while (n--)
		v = (v << 1) | br_bit(b);
	return v;
}

static unsigned br_peek(br *b, int n)
{
	int save = b->pos, e = b->err;
	unsigned v = 0;
	int i;

	for (i = 0; i < n; i++) {
		unsigned bit = 0;

		if (b->pos < b->nbits + 1)
			bit = (b->p[b->pos >> 3] >> (7 - (b->pos & 7))) & 1;
		b->pos++;
		v = (v << 1) | bit;
	}
	b->pos = save;
	b->err = e;
	return v;
}

static unsigned br_ue(br *b)
{
	int z = 0;

	while (!br_bit(b)) {
		if (b->err || ++z > 31) {
			b->err = 1;
			return 0;
		}
	}
	return (1u << z) - 1 + br_u(b, z);
}

static int br_se(br *b)
{
	unsigned k = br_ue(b);

	return (k & 1) ? (int)((k + 1) >> 1) : -(int)(k >> 1);
}

static int br_more(const br *b)
{
	return b->pos < b->nbits;
}

/* ---------------- helpers ---------------- */
static int fail(h264o_dec *d, const char *msg)
{
	snprintf(d->err, sizeof(d->err), "%s", msg);
	return -1;
}

static int clip8(int v)
{
	return v < 0 ? 0 : v > 255 ? 255 : v;
}

static int clip3(int lo, int hi, int v)
{
	return v < lo ? lo : v > hi ? hi : v;
}

h264o_dec *h264o_dec_open(void)
{
	return calloc(1, sizeof(h264o_dec));
}

void h264o_dec_close(h264o_dec *d)
{
	if (!d)
		return;
	free(d->cur_y);
	free(d->cur_uv);
	free(d->ref_y);
	free(d->ref_uv);
	free(d->mbs);
	free(d->slice_idc);
	free(d->slice_a);
	free(d->slice_b);
	free(d->rbsp);
	free(d);
}

int h264o_dec_width(const h264o_dec *d) { return d->cw; }
int h264o_dec_height(const h264o_dec *d) { return d->ch; }
int h264o_dec_crop_width(const h264o_dec *d) { return d->cw - 2 * d->crop_r; }
int h264o_dec_crop_height(const h264o_dec *d) { return d->ch - 2 * d->crop_b; }
const uint8_t *h264o_dec_luma(const h264o_dec *d) { return d->ref_y; }
const uint8_t *h264o_dec_chroma(const h264o_dec *d) { return d->ref_uv; }
const char *h264o_dec_error(const h264o_dec *d) { return d->err; }
/* frame rate from the VUI as num/den (0/0 when the SPS has no timing) */
void h264o_dec_frame_rate(const h264o_dec *d, unsigned *num, unsigned *den)
{
	*num = d->vui_ticks ? d->vui_scale : 0;
	*den = d->vui_ticks * 2;
}
int h264o_dec_reorder_frames(const h264o_dec *d) { return d->vui_reorder; }
int h264o_dec_aud_count(const h264o_dec *d) { return d->n_aud; }

/* ---------------- parameter sets ---------------- */
/* vui_parameters (H.264 E.1.1): timing and the bitstream restriction are kept, the rest skipped */
static int parse_vui(h264o_dec *d, br *b)
{
	if (br_u(b, 1)) { /* aspect_ratio_info */
		if (br_u(b, 8) == 255)
			br_u(b, 32);
	}
	if (br_u(b, 1)) /* overscan_info */
		br_u(b, 1);
	if (br_u(b, 1)) { /* video_signal_type */
		br_u(b, 4);
		if (br_u(b, 1))
			br_u(b, 24);
	}
	if (br_u(b, 1)) { /* chroma_loc_info */
		br_ue(b);
		br_ue(b);
	}
	if (br_u(b, 1)) { /* timing_info */
		d->vui_ticks = br_u(b, 32);
		d->vui_scale = br_u(b, 32);
		br_u(b, 1);
	}
	if (br_u(b, 1) || br_u(b, 1))
		return fail(d, "HRD parameters unsupported");
	br_u(b, 1); /* pic_struct_present_flag */
	if (br_u(b, 1)) { /* bitstream_restriction */
		br_u(b, 1);
		br_ue(b);
		br_ue(b);
		br_ue(b);
		br_ue(b);
		d->vui_reorder = (int)br_ue(b);
		br_ue(b);
	}
	return 0;
}

static int parse_sps(h264o_dec *d, br *b)
{
	int profile = (int)br_u(b, 8), mbw, mbh;

	br_u(b, 8);
	br_u(b, 8);
	if (br_ue(b) != 0)
		return fail(d, "sps id != 0");
	if (profile >= 100)
		return fail(d, "high profiles unsupported");
	d->log2_max_frame_num = (int)br_ue(b) + 4;
	d->poc_type = (int)br_ue(b);
	if (d->poc_type == 0)
		br_ue(b);
	else if (d->poc_type == 1)
		return fail(d, "poc type 1 unsupported");
	br_ue(b); /* max_num_ref_frames */
	br_u(b, 1);
	mbw = (int)br_ue(b) + 1;
	mbh = (int)br_ue(b) + 1;
	if (!br_u(b, 1))
		return fail(d, "interlace unsupported");
	br_u(b, 1);
	d->crop_r = d->crop_b = 0;
	if (br_u(b, 1)) {
		if (br_ue(b) != 0)
			return fail(d, "crop left unsupported");
		d->crop_r = (int)br_ue(b);
		if (br_ue(b) != 0)
			return fail(d, "crop top unsupported");
		d->crop_b = (int)br_ue(b);
	}
	d->vui_ticks = d->vui_scale = 0;
	d->vui_reorder = -1;
	if (br_u(b, 1) && parse_vui(d, b) < 0)
		return -1;
	if (b->err)
		return fail(d, "sps truncated");
	if (!d->have_sps || mbw != d->mb_w || mbh != d->mb_h) {
		size_t n = (size_t)mbw * mbh;

		free(d->cur_y); free(d->cur_uv); free(d->ref_y); free(d->ref_uv); free(d->mbs);
		free(d->slice_idc); free(d->slice_a); free(d->slice_b);
		d->mb_w = mbw;
		d->mb_h = mbh;
		d->cw = mbw * 16;
		d->ch = mbh * 16;
		d->cur_y = calloc(n, 256);
		d->ref_y = calloc(n, 256);
		d->cur_uv = calloc(n, 128);
		d->ref_uv = calloc(n, 128);
		d->mbs = calloc(n, sizeof(dmb));
		d->slice_idc = calloc(n, sizeof(int));
		d->slice_a = calloc(n, sizeof(int));
		d->slice_b = calloc(n, sizeof(int));
		if (!d->cur_y || !d->ref_y || !d->cur_uv || !d->ref_uv || !d->mbs)
			return fail(d, "out of memory");
		d->have_ref = 0;
	}
	d->have_sps = 1;
	return 0;
}

static int parse_pps(h264o_dec *d, br *b)
{
	if (br_ue(b) != 0 || br_ue(b) != 0)
		return fail(d, "pps/sps id != 0");
	if (br_u(b, 1))
		return fail(d, "CABAC unsupported");
	br_u(b, 1);
	if (br_ue(b) != 0)
		return fail(d, "slice groups unsupported");
	if (br_ue(b) != 0)
		return fail(d, "num_ref_idx_l0 > 1 unsupported");
	br_ue(b);
	if (br_u(b, 1) || br_u(b, 2))
		return fail(d, "weighted prediction unsupported");
	d->init_qp = 26 + br_se(b);
	br_se(b);
	d->chroma_qp_offset = br_se(b);
	d->deblock_ctrl = (int)br_u(b, 1);
	d->constrained_intra = (int)br_u(b, 1);
	if (br_u(b, 1))
		return fail(d, "redundant pictures unsupported");
	if (b->err)
		return fail(d, "pps truncated");
	d->have_pps = 1;
	return 0;
}

/* ---------------- CAVLC residual ---------------- */
static int read_vlc(br *b, const uint8_t *lens, const uint8_t *bits, int n)
{
	int i;

	for (i = 0; i < n; i++)
		if (lens[i] && br_peek(b, lens[i]) == bits[i]) {
			b->pos += lens[i];
			return i;
		}
	b->err = 1;
	return -1;
}

/* decode one block into lv[0..maxc-1] (zig-zag order); returns total_coeff */
static int residual_block(br *b, int16_t *lv, int maxc, int nC)
{
	int tok, total, t1, i, suffix_len, zeros_left;
	int level[16], run[16];

	memset(lv, 0, sizeof(int16_t) * (size_t)maxc);
	if (nC < 0)
		tok = read_vlc(b, h264o_chroma_dc_coeff_token_len, h264o_chroma_dc_coeff_token_bits, 20);
	else {
		int tab = nC < 2 ? 0 : nC < 4 ? 1 : nC < 8 ? 2 : 3;

		tok = read_vlc(b, h264o_coeff_token_len[tab], h264o_coeff_token_bits[tab], 68);
	}
	if (tok < 0)
		return -1;
	total = tok >> 2;
	t1 = tok & 3;
	if (total > maxc) {
		b->err = 1;
		return -1;
	}
	if (!total)
		return 0;
	suffix_len = (total > 10 && t1 < 3) ? 1 : 0;
	for (i = 0; i < total; i++) {
		if (i < t1) {
			level[i] = br_u(b, 1) ? -1 : 1;
		} else {
			int prefix = 0, code, size;

			while (!br_bit(b)) {
				if (b->err || ++prefix > 15) {
					b->err = 1;
					return -1;
				}
			}
			code = (prefix < 15 ? prefix : 15) << suffix_len;
			if (prefix == 14 && suffix_len == 0)
				size = 4;
			else if (prefix >= 15)
				size = prefix - 3;
			else
				size = suffix_len;
			if (size)
				code += (int)br_u(b, size);
			if (prefix >= 15 && suffix_len == 0)
				code += 15;
			if (i == t1 && t1 < 3)
				code += 2;
			level[i] = (code & 1) ? (-code - 1) >> 1 : (code + 2) >> 1;
			if (suffix_len == 0)
				suffix_len = 1;
			if (abs(level[i]) > (3 << (suffix_len - 1)) && suffix_len < 6)
				suffix_len++;
		}
	}
	if (total < maxc) {
		if (maxc == 4)
			zeros_left = read_vlc(b, h264o_chroma_dc_total_zeros_len[total - 1],
					      h264o_chroma_dc_total_zeros_bits[total - 1], 4);
		else
			zeros_left = read_vlc(b, h264o_total_zeros_len[total - 1],
					      h264o_total_zeros_bits[total - 1], 16);
		if (zeros_left < 0)
			return -1;
	} else {
		zeros_left = 0;
	}
	for (i = 0; i < total - 1; i++) {
		if (zeros_left > 0) {
			int t = (zeros_left > 7 ? 7 : zeros_left) - 1;

			run[i] = read_vlc(b, h264o_run_before_len[t], h264o_run_before_bits[t], 15);
			if (run[i] < 0)
				return -1;
		} else {
			run[i] = 0;
		}
		zeros_left -= run[i];
		if (zeros_left < 0) {
			b->err = 1;
			return -1;
		}
	}
	run[total - 1] = zeros_left;
	{
		int pos = -1;

		for (i = total - 1; i >= 0; i--) {
			pos += run[i] + 1;
			if (pos >= maxc) {
				b->err = 1;
				return -1;
			}
			lv[pos] = (int16_t)level[i];
		}
	}
	return total;
}

/* ---------------- neighbour access ---------------- */
static const dmb *mb_at(const h264o_dec *d, int mx, int my, int cur_slice, int cur_addr)
{
	int addr;

	if (mx < 0 || my < 0 || mx >= d->mb_w || my >= d->mb_h)
		return NULL;
	addr = my * d->mb_w + mx;
	if (addr >= cur_addr || d->mbs[addr].slice != cur_slice)
		return NULL;
	return &d->mbs[addr];
}

/* the neighbour as INTRA PREDICTION sees it (8.3.1.2, 8.3.3, 8.3.4): with constrained_intra_pred_flag
 * a macroblock coded in Inter prediction mode counts as not available; the same rule decides
 * dcPredModePredictedFlag (8.3.1.1). CAVLC's nC keeps looking at the real neighbour (9.2.1: the
 * exception there is for data partitioning only). */
static const dmb *mb_at_intra(const h264o_dec *d, int mx, int my, int cur_slice, int cur_addr)
{
	const dmb *n = mb_at(d, mx, my, cur_slice, cur_addr);

	if (n && d->constrained_intra && n->type >= H264O_MB_P16x16)
		return NULL;
	return n;
}

static int nc_of(const h264o_dec *d, const dmb *cur, const dmb *A, const dmb *B,
		 int plane, int bx, int by)
{
	int w = plane ? 2 : 4, base = plane ? 16 + 4 * (plane - 1) : 0;
	int na = -1, nb = -1;

	(void)d;
	if (bx > 0)
		na = cur->nnz[base + bx - 1 + w * by];
	else if (A)
		na = A->nnz[base + w - 1 + w * by];
	if (by > 0)
		nb = cur->nnz[base + bx + w * (by - 1)];
	else if (B)
		nb = B->nnz[base + bx + w * (w - 1)];
	if (na >= 0 && nb >= 0)
		return (na + nb + 1) >> 1;
	return na >= 0 ? na : nb >= 0 ? nb : 0;
}

/* ---------------- reconstruction primitives ---------------- */
/* inverse transform of scaled coefficients c (raster [row][col]) per 8.5.12.2,
 * added to the prediction already stored at dst */
static void inv4x4_add(const int c[4][4], uint8_t *dst, int stride)
{
	int f[4][4], i, j;

	for (i = 0; i < 4; i++) { /* each row */
		int e0 = c[i][0] + c[i][2], e1 = c[i][0] - c[i][2];
		int e2 = (c[i][1] >> 1) - c[i][3], e3 = c[i][1] + (c[i][3] >> 1);

		f[i][0] = e0 + e3;
		f[i][1] = e1 + e2;
		f[i][2] = e1 - e2;
		f[i][3] = e0 - e3;
	}
	for (j = 0; j < 4; j++) { /* each column */
		int g0 = f[0][j] + f[2][j], g1 = f[0][j] - f[2][j];
		int g2 = (f[1][j] >> 1) - f[3][j], g3 = f[1][j] + (f[3][j] >> 1);
		int h[4];

		h[0] = g0 + g3;
		h[1] = g1 + g2;
		h[2] = g1 - g2;
		h[3] = g0 - g3;
		for (i = 0; i < 4; i++)
			dst[i * stride + j] = (uint8_t)clip8(dst[i * stride + j] + ((h[i] + 32) >> 6));
	}
}

static int level_scale(int qp, int i, int j)
{
	int m = qp % 6;

	if (!(i & 1) && !(j & 1))
		return 16 * h264o_dequant_v[m][0];
	if ((i & 1) && (j & 1))
		return 16 * h264o_dequant_v[m][1];
	return 16 * h264o_dequant_v[m][2];
}

/* scale a zig-zag level list into c[][]; first = 1 keeps c[0][0] untouched */
static void scale4x4(const int16_t *lv, int qp, int first, int c[4][4])
{
	int k;

	for (k = first; k < 16; k++) {
		int r = h264o_zigzag4x4[k], i = r >> 2, j = r & 3;
		int ls = level_scale(qp, i, j);

		if (qp >= 24)
			c[i][j] = (lv[k] * ls) * (1 << (qp / 6 - 4));
		else
			c[i][j] = (lv[k] * ls + (1 << (3 - qp / 6))) >> (4 - qp / 6);
	}
}

/* Intra 4x4 prediction straight from the picture with the spec's p[x,y] */
static void intra4x4(uint8_t *pic, int stride, int x0, int y0, int mode,
		     int top, int left, int tr)
{
	int p[13 + 8]; /* P(x,-1) x=-1..7 at [x+1], P(-1,y) y=0..3 at [9+y] */
	int x, y;
	uint8_t *dst = pic + y0 * stride + x0;
#define PT(xx) p[(xx) + 1]
#define PL(yy) p[9 + (yy)]
	for (x = 0; x < 21; x++)
		p[x] = 128;
	if (top) {
		for (x = 0; x < 4; x++)
			PT(x) = pic[(y0 - 1) * stride + x0 + x];
		for (x = 4; x < 8; x++)
			PT(x) = tr ? pic[(y0 - 1) * stride + x0 + x] : PT(3);
	}
	if (left)
		for (y = 0; y < 4; y++)
			PL(y) = pic[(y0 + y) * stride + x0 - 1];
	if (top && left)
		PT(-1) = pic[(y0 - 1) * stride + x0 - 1];
	for (y = 0; y < 4; y++)
		for (x = 0; x < 4; x++) {
			int v;

			switch (mode) {
			case 0: v = PT(x); break;
			case 1: v = PL(y); break;
			case 2:
				if (top && left)
					v = (PT(0) + PT(1) + PT(2) + PT(3) + PL(0) + PL(1) + PL(2) + PL(3) + 4) >> 3;
				else if (left)
					v = (PL(0) + PL(1) + PL(2) + PL(3) + 2) >> 2;
				else if (top)
					v = (PT(0) + PT(1) + PT(2) + PT(3) + 2) >> 2;
				else
					v = 128;
				break;
			case 3:
				if (x == 3 && y == 3)
					v = (PT(6) + 3 * PT(7) + 2) >> 2;
				else
					v = (PT(x + y) + 2 * PT(x + y + 1) + PT(x + y + 2) + 2) >> 2;
				break;
			case 4:
				if (x > y)
					v = (PT(x - y - 2) + 2 * PT(x - y - 1) + PT(x - y) + 2) >> 2;
				else if (x < y)
					v = ((y - x - 2 >= 0 ? PL(y - x - 2) : PT(-1)) + 2 * PL(y - x - 1) + PL(y - x) + 2) >> 2;
				else
					v = (PT(0) + 2 * PT(-1) + PL(0) + 2) >> 2;
				break;
			case 5: {
				int z = 2 * x - y;

				if (z >= 0 && !(z & 1))
					v = (PT(x - (y >> 1) - 1) + PT(x - (y >> 1)) + 1) >> 1;
				else if (z >= 0)
					v = (PT(x - (y >> 1) - 2) + 2 * PT(x - (y >> 1) - 1) + PT(x - (y >> 1)) + 2) >> 2;
				else if (z == -1)
					v = (PL(0) + 2 * PT(-1) + PT(0) + 2) >> 2;
				else
					v = (PL(y - 1) + 2 * PL(y - 2) + (y - 3 >= 0 ? PL(y - 3) : PT(-1)) + 2) >> 2;
				break;
			}
			case 6: {
				int z = 2 * y - x;
#define PLN(k) ((k) >= 0 ? PL(k) : PT(-1))
				if (z >= 0 && !(z & 1))
					v = (PLN(y - (x >> 1) - 1) + PL(y - (x >> 1)) + 1) >> 1;
				else if (z >= 0)
					v = (PLN(y - (x >> 1) - 2) + 2 * PLN(y - (x >> 1) - 1) + PL(y - (x >> 1)) + 2) >> 2;
				else if (z == -1)
					v = (PL(0) + 2 * PT(-1) + PT(0) + 2) >> 2;
				else
					v = (PT(x - 1) + 2 * PT(x - 2) + PT(x - 3) + 2) >> 2;
#undef PLN
				break;
			}
			case 7:
				if (!(y & 1))
					v = (PT(x + (y >> 1)) + PT(x + (y >> 1) + 1) + 1) >> 1;
				else
					v = (PT(x + (y >> 1)) + 2 * PT(x + (y >> 1) + 1) + PT(x + (y >> 1) + 2) + 2) >> 2;
				break;
			default: {
				int z = x + 2 * y;

				if (z > 5)
					v = PL(3);
				else if (z == 5)
					v = (PL(2) + 3 * PL(3) + 2) >> 2;
				else if (!(z & 1))
					v = (PL(y + (x >> 1)) + PL(y + (x >> 1) + 1) + 1) >> 1;
				else
					v = (PL(y + (x >> 1)) + 2 * PL(y + (x >> 1) + 1) + PL(y + (x >> 1) + 2) + 2) >> 2;
				break;
			}
			}
			dst[y * stride + x] = (uint8_t)v;
		}
#undef PT
#undef PL
}

/* generic NxN plane/DC/H/V predictor on a sample-step grid: luma (n=16,
 * step=1) and nv12 chroma (n=8, step=2). mode: 0 V 1 H 2 DC 3 Plane */
static void intra_big(uint8_t *base, int stride, int step, int n, int mode, int top, int left)
{
	int x, y;
#define S(xx, yy) base[(yy) * stride + (xx) * step]
	if (mode == 0) {
		for (y = 0; y < n; y++)
			for (x = 0; x < n; x++)
				S(x, y) = S(x, -1);
	} else if (mode == 1) {
		for (y = 0; y < n; y++)
			for (x = 0; x < n; x++)
				S(x, y) = S(-1, y);
	} else if (mode == 3) {
		int h = 0, v = 0, a, b, c, half = n / 2, k = n == 16 ? 5 : 34;

		for (x = 0; x < half; x++)
			h += (x + 1) * (S(half + x, -1) - S(half - 2 - x, -1));
		for (y = 0; y < half; y++)
			v += (y + 1) * (S(-1, half + y) - S(-1, half - 2 - y));
		a = 16 * (S(-1, n - 1) + S(n - 1, -1));
		b = (k * h + 32) >> 6;
		c = (k * v + 32) >> 6;
		for (y = 0; y < n; y++)
			for (x = 0; x < n; x++)
				S(x, y) = (uint8_t)clip8((a + b * (x - (half - 1)) + c * (y - (half - 1)) + 16) >> 5);
	} else if (n == 16) {
		int s = 0, v;

		if (top)
			for (x = 0; x < 16; x++)
				s += S(x, -1);
		if (left)
			for (y = 0; y < 16; y++)
				s += S(-1, y);
		v = top && left ? (s + 16) >> 5 : (top || left) ? (s + 8) >> 4 : 128;
		for (y = 0; y < 16; y++)
			for (x = 0; x < 16; x++)
				S(x, y) = (uint8_t)v;
	} else {
		int bx, by;

		for (by = 0; by < 2; by++)
			for (bx = 0; bx < 2; bx++) {
				int st = 0, sl = 0, v, use_t = top, use_l = left;

				if (bx == 1 && by == 0 && top)
					use_l = 0;
				if (bx == 0 && by == 1 && left)
					use_t = 0;
				if (use_t)
					for (x = 0; x < 4; x++)
						st += S(4 * bx + x, -1);
				if (use_l)
					for (y = 0; y < 4; y++)
						sl += S(-1, 4 * by + y);
				if (use_t && use_l)
					v = (st + sl + 4) >> 3;
				else if (use_t)
					v = (st + 2) >> 2;
				else if (use_l)
					v = (sl + 2) >> 2;
				else
					v = 128;
				for (y = 0; y < 4; y++)
					for (x = 0; x < 4; x++)
						S(4 * bx + x, 4 * by + y) = (uint8_t)v;
			}
	}
#undef S
}

/* ---------------- macroblock decode ---------------- */
typedef struct slice_ctx {
	int type;   /* 0 P, 2 I */
	int qp;
	int num;    /* slice number in picture */
} slice_ctx;

static int chroma_qp(const h264o_dec *d, int qp)
{
	return h264o_chroma_qp[clip3(0, 51, qp + d->chroma_qp_offset)];
}

static int decode_residual_and_recon(h264o_dec *d, br *b, dmb *m, int mx, int my, int addr,
				     int cbp, int is16, int16_t dc16[16])
{
	const dmb *A = mb_at(d, mx - 1, my, m->slice, addr), *B = mb_at(d, mx, my - 1, m->slice, addr);
	int qp = m->qp, qpc = chroma_qp(d, qp);
	int i8, i4, c, blk, k;
	int16_t lv[16];
	int dcy[4][4];

	memset(dcy, 0, sizeof(dcy));
	if (is16) {
		int cm[4][4], f[4][4], g[4][4], i, j;

		if (residual_block(b, dc16, 16, nc_of(d, m, A, B, 0, 0, 0)) < 0)
			return fail(d, "bad Intra16x16 DC block");
		for (k = 0; k < 16; k++)
			cm[h264o_zigzag4x4[k] >> 2][h264o_zigzag4x4[k] & 3] = dc16[k];
		for (i = 0; i < 4; i++) {
			f[i][0] = cm[i][0] + cm[i][1] + cm[i][2] + cm[i][3];
			f[i][1] = cm[i][0] + cm[i][1] - cm[i][2] - cm[i][3];
			f[i][2] = cm[i][0] - cm[i][1] - cm[i][2] + cm[i][3];
			f[i][3] = cm[i][0] - cm[i][1] + cm[i][2] - cm[i][3];
		}
		for (j = 0; j < 4; j++) {
			g[0][j] = f[0][j] + f[1][j] + f[2][j] + f[3][j];
			g[1][j] = f[0][j] + f[1][j] - f[2][j] - f[3][j];
			g[2][j] = f[0][j] - f[1][j] - f[2][j] + f[3][j];
			g[3][j] = f[0][j] - f[1][j] + f[2][j] - f[3][j];
		}
		for (i = 0; i < 4; i++)
			for (j = 0; j < 4; j++) {
				int ls = level_scale(qp, 0, 0);

				if (qp >= 36)
					dcy[i][j] = (g[i][j] * ls) * (1 << (qp / 6 - 6));
				else
					dcy[i][j] = (g[i][j] * ls + (1 << (5 - qp / 6))) >> (6 - qp / 6);
			}
	}
	for (i8 = 0; i8 < 4; i8++)
		for (i4 = 0; i4 < 4; i4++) {
			int idx = 4 * i8 + i4, bx = h264o_blk_x[idx], by = h264o_blk_y[idx];
			int c4[4][4];
			int n = 0;

			memset(c4, 0, sizeof(c4));
			memset(lv, 0, sizeof(lv));
			if (cbp & (1 << i8)) {
				int nC = nc_of(d, m, A, B, 0, bx, by);

				if (is16)
					n = residual_block(b, lv + 1, 15, nC);
				else
					n = residual_block(b, lv, 16, nC);
				if (n < 0)
					return fail(d, "bad luma block");
			}
			m->nnz[bx + 4 * by] = (uint8_t)n;
			if (is16) {
				scale4x4(lv, qp, 1, c4);
				c4[0][0] = dcy[by][bx];
			} else if (m->type == H264O_MB_I4x4) {
				/* prediction happens block by block for I4x4: done by caller order */
				scale4x4(lv, qp, 0, c4);
			} else {
				scale4x4(lv, qp, 0, c4);
			}
			if (m->type == H264O_MB_I4x4) {
				const dmb *Ai = mb_at_intra(d, mx - 1, my, m->slice, addr), *Bi = mb_at_intra(d, mx, my - 1, m->slice, addr);
				int top = by > 0 || Bi, left = bx > 0 || Ai, tr;
				static const uint8_t tri[16] = { 0, 0, 1, 0, 0, 0, 1, 0, 1, 1, 1, 0, 1, 0, 1, 0 };

				if (by == 0)
					tr = bx < 3 ? Bi != NULL : mb_at_intra(d, mx + 1, my - 1, m->slice, addr) != NULL;
				else
					tr = tri[idx];
				/* 8.3.1.2: every mode but DC is only allowed with the samples it reads (modes 4..6 read the
				 * corner: of macroblock D for block 0, of A / B along the macroblock's edges) */
				{
					const int mode = m->i4[bx + 4 * by];
					const int need_t = mode == 0 || mode == 3 || mode == 7 || (mode >= 4 && mode <= 6);
					const int need_l = mode == 1 || mode == 8 || (mode >= 4 && mode <= 6);

					if ((need_t && !top) || (need_l && !left))
						return fail(d, "Intra4x4 mode needs an unavailable neighbour");
				}
				if (idx == 0 && m->i4[0] >= 4 && m->i4[0] <= 6 && !mb_at_intra(d, mx - 1, my - 1, m->slice, addr))
					return fail(d, "Intra4x4 mode needs an unavailable corner");
				intra4x4(d->cur_y, d->cw, mx * 16 + 4 * bx, my * 16 + 4 * by, m->i4[bx + 4 * by],
					 top, left, tr);
			}
			inv4x4_add(c4, d->cur_y + (my * 16 + 4 * by) * d->cw + mx * 16 + 4 * bx, d->cw);
		}
	/* chroma */
	{
		int16_t cdc[2][4];
		int dcc[2][4];

		memset(cdc, 0, sizeof(cdc));
		if (cbp & 0x30)
			for (c = 0; c < 2; c++)
				if (residual_block(b, cdc[c], 4, -1) < 0)
					return fail(d, "bad chroma DC block");
		for (c = 0; c < 2; c++) {
			int f0 = cdc[c][0] + cdc[c][1] + cdc[c][2] + cdc[c][3];
			int f1 = cdc[c][0] - cdc[c][1] + cdc[c][2] - cdc[c][3];
			int f2 = cdc[c][0] + cdc[c][1] - cdc[c][2] - cdc[c][3];
			int f3 = cdc[c][0] - cdc[c][1] - cdc[c][2] + cdc[c][3];
			int ls = level_scale(qpc, 0, 0);

			dcc[c][0] = ((f0 * ls) * (1 << (qpc / 6))) >> 5;
			dcc[c][1] = ((f1 * ls) * (1 << (qpc / 6))) >> 5;
			dcc[c][2] = ((f2 * ls) * (1 << (qpc / 6))) >> 5;
			dcc[c][3] = ((f3 * ls) * (1 << (qpc / 6))) >> 5;
		}
		for (c = 0; c < 2; c++)
			for (blk = 0; blk < 4; blk++) {
				int bx = blk & 1, by = blk >> 1, n = 0, x, y;
				int c4[4][4];
				uint8_t tmp[16];

				memset(c4, 0, sizeof(c4));
				memset(lv, 0, sizeof(lv));
				if (cbp & 0x20) {
					n = residual_block(b, lv + 1, 15, nc_of(d, m, A, B, 1 + c, bx, by));
					if (n < 0)
						return fail(d, "bad chroma AC block");
				}
				m->nnz[16 + 4 * c + blk] = (uint8_t)n;
				scale4x4(lv, qpc, 1, c4);
				c4[0][0] = dcc[c][blk];
				for (y = 0; y < 4; y++)
					for (x = 0; x < 4; x++)
						tmp[4 * y + x] = d->cur_uv[(my * 8 + 4 * by + y) * d->cw + 2 * (mx * 8 + 4 * bx + x) + c];
				inv4x4_add(c4, tmp, 4);
				for (y = 0; y < 4; y++)
					for (x = 0; x < 4; x++)
						d->cur_uv[(my * 8 + 4 * by + y) * d->cw + 2 * (mx * 8 + 4 * bx + x) + c] = tmp[4 * y + x];
			}
	}
	return b->err ? fail(d, "residual truncated") : 0;
}

static int decode_intra_mb(h264o_dec *d, br *b, dmb *m, int mx, int my, int addr, int mbt)
{
	/* everything in this function is intra prediction: the constrained view of the neighbours */
	const dmb *A = mb_at_intra(d, mx - 1, my, m->slice, addr), *B = mb_at_intra(d, mx, my - 1, m->slice, addr);
	int cbp, cmode, c;
	int16_t dc16[16];

	m->mvx = m->mvy = 0;
	memset(m->nnz, 0, sizeof(m->nnz));
	if (mbt == 0) {
		int idx;

		m->type = H264O_MB_I4x4;
		for (idx = 0; idx < 16; idx++) {
			int bx = h264o_blk_x[idx], by = h264o_blk_y[idx];
			int ma = 2, mb_ = 2, pm, dcflag = 0;

			if (bx > 0)
				ma = m->i4[bx - 1 + 4 * by];
			else if (A)
				ma = A->type == H264O_MB_I4x4 ? A->i4[3 + 4 * by] : 2;
			else
				dcflag = 1;
			if (by > 0)
				mb_ = m->i4[bx + 4 * (by - 1)];
			else if (B)
				mb_ = B->type == H264O_MB_I4x4 ? B->i4[bx + 12] : 2;
			else
				dcflag = 1;
			pm = dcflag ? 2 : (ma < mb_ ? ma : mb_);
			if (br_u(b, 1))
				m->i4[bx + 4 * by] = (int8_t)pm;
			else {
				int rem = (int)br_u(b, 3);

				m->i4[bx + 4 * by] = (int8_t)(rem < pm ? rem : rem + 1);
			}
		}
		cmode = (int)br_ue(b);
		{
			unsigned code = br_ue(b);

			if (code > 47)
				return fail(d, "bad coded_block_pattern");
			cbp = h264o_cbp_intra_from_code[code];
		}
		if (cbp) {
			int dq = br_se(b);

			if (dq < -26 || dq > 25)
				return fail(d, "mb_qp_delta out of range");
			m->qp = (int8_t)((m->qp + dq + 52) % 52);
		}
	} else if (mbt <= 24) {
		int t = mbt - 1;

		m->type = H264O_MB_I16x16;
		memset(m->i4, 2, sizeof(m->i4));
		cbp = ((t / 12) ? 15 : 0) | (((t / 4) % 3) << 4);
		cmode = (int)br_ue(b);
		{
			int dq = br_se(b);

			if (dq < -26 || dq > 25)
				return fail(d, "mb_qp_delta out of range");
			m->qp = (int8_t)((m->qp + dq + 52) % 52);
		}
		/* luma prediction for the whole MB */
		if (((t & 3) == 0 && !B) || ((t & 3) == 1 && !A) ||
		    ((t & 3) == 3 && !(A && B && mb_at_intra(d, mx - 1, my - 1, m->slice, addr))))
			return fail(d, "Intra16x16 mode needs an unavailable neighbour");
		intra_big(d->cur_y + my * 16 * d->cw + mx * 16, d->cw, 1, 16, t & 3, B != NULL, A != NULL);
	} else if (mbt == 25) {
		/* I_PCM (7.3.5, 8.3.5): pcm_alignment_zero_bits, then the samples themselves; every
		 * block counts 16 coefficients for its neighbours' nC (9.2.1) */
		int x, y;

		m->type = H264O_MB_IPCM;
		memset(m->i4, 2, sizeof(m->i4));
		memset(m->nnz, 16, sizeof(m->nnz));
		/* (no mb_qp_delta: QP_Y stays what it was, 7.4.5; only deblocking takes it as 0, 8.7.2.2: dmb_qp) */
		while (b->pos & 7)
			if (br_u(b, 1))
				return fail(d, "pcm_alignment_zero_bit is 1");
		for (y = 0; y < 16; y++)
			for (x = 0; x < 16; x++)
				d->cur_y[(my * 16 + y) * d->cw + mx * 16 + x] = (uint8_t)br_u(b, 8);
		for (c = 0; c < 2; c++)
			for (y = 0; y < 8; y++)
				for (x = 0; x < 8; x++)
					d->cur_uv[(my * 8 + y) * d->cw + 2 * (mx * 8 + x) + c] = (uint8_t)br_u(b, 8);
		return b->err ? fail(d, "I_PCM samples truncated") : 0;
	} else {
		return fail(d, "bad mb_type");
	}
	if (cmode > 3 || (cmode == 1 && !A) || (cmode == 2 && !B) ||
	    (cmode == 3 && !(A && B && mb_at_intra(d, mx - 1, my - 1, m->slice, addr))))
		return fail(d, "bad intra_chroma_pred_mode");
	/* chroma prediction: syntax 0 DC 1 H 2 V 3 Plane -> intra_big 2,1,0,3 */
	for (c = 0; c < 2; c++)
		intra_big(d->cur_uv + my * 8 * d->cw + 2 * mx * 8 + c, d->cw, 2, 8,
			  cmode == 0 ? 2 : cmode == 1 ? 1 : cmode == 2 ? 0 : 3, B != NULL, A != NULL);
	return decode_residual_and_recon(d, b, m, mx, my, addr, cbp, m->type == H264O_MB_I16x16, dc16);
}

static int med3(int a, int b, int c)
{
	int mn = a < b ? a : b, mx = a < b ? b : a;

	return c < mn ? mn : c > mx ? mx : c;
}

/* mv prediction for a 16x16 partition; neighbours: inter with refIdx 0, or
 * unavailable/intra (refIdx -1). skip = 1 returns the P_Skip vector. */
static void pred_mv(const h264o_dec *d, int mx, int my, int slice, int addr, int skip, int *px, int *py)
{
	const dmb *A = mb_at(d, mx - 1, my, slice, addr), *B = mb_at(d, mx, my - 1, slice, addr);
	const dmb *C = mb_at(d, mx + 1, my - 1, slice, addr);
	int ra, rb, rc, ax = 0, ay = 0, bx = 0, by = 0, cx = 0, cy = 0;
	int c_avail = C != NULL;

	if (!C) {
		C = mb_at(d, mx - 1, my - 1, slice, addr);
		c_avail = C != NULL;
	}
	ra = A && A->type >= H264O_MB_P16x16 ? 0 : -1;
	rb = B && B->type >= H264O_MB_P16x16 ? 0 : -1;
	rc = C && C->type >= H264O_MB_P16x16 ? 0 : -1;
	if (ra == 0) { ax = A->mvx; ay = A->mvy; }
	if (rb == 0) { bx = B->mvx; by = B->mvy; }
	if (rc == 0) { cx = C->mvx; cy = C->mvy; }
	if (skip && (!A || !B || (ra == 0 && ax == 0 && ay == 0) || (rb == 0 && bx == 0 && by == 0))) {
		*px = *py = 0;
		return;
	}
	if (!B && !c_avail && A) {
		*px = ax; *py = ay; /* B, C := A */
		return;
	}
	if ((ra == 0) + (rb == 0) + (rc == 0) == 1) {
		*px = ra == 0 ? ax : rb == 0 ? bx : cx;
		*py = ra == 0 ? ay : rb == 0 ? by : cy;
		return;
	}
	*px = med3(ax, bx, cx);
	*py = med3(ay, by, cy);
}


/* clause 8.4.2.2.1: the sample at integer position (x, y) + quarter fractions (fx, fy) of the reference
 * picture; positions outside the picture read the nearest edge sample */
static int dec_ref(const h264o_dec *d, int x, int y)
{
	return d->ref_y[clip3(0, d->ch - 1, y) * d->cw + clip3(0, d->cw - 1, x)];
}

static int dec_tap(int a, int b, int c, int dd, int e, int f) { return a - 5 * b + 20 * c + 20 * dd - 5 * e + f; }

static int dec_b1(const h264o_dec *d, int x, int y)
{
	return dec_tap(dec_ref(d, x - 2, y), dec_ref(d, x - 1, y), dec_ref(d, x, y), dec_ref(d, x + 1, y), dec_ref(d, x + 2, y), dec_ref(d, x + 3, y));
}

static int dec_h1(const h264o_dec *d, int x, int y)
{
	return dec_tap(dec_ref(d, x, y - 2), dec_ref(d, x, y - 1), dec_ref(d, x, y), dec_ref(d, x, y + 1), dec_ref(d, x, y + 2), dec_ref(d, x, y + 3));
}

static int dec_luma_sample(const h264o_dec *d, int x, int y, int fx, int fy)
{
	int g = dec_ref(d, x, y), b, h, s, m, j;

	if (!fx && !fy)
		return g;
	b = clip3(0, 255, (dec_b1(d, x, y) + 16) >> 5);
	h = clip3(0, 255, (dec_h1(d, x, y) + 16) >> 5);
	if (!fy)
		return fx == 2 ? b : fx == 1 ? (g + b + 1) >> 1 : (dec_ref(d, x + 1, y) + b + 1) >> 1;
	if (!fx)
		return fy == 2 ? h : fy == 1 ? (g + h + 1) >> 1 : (dec_ref(d, x, y + 1) + h + 1) >> 1;
	s = clip3(0, 255, (dec_b1(d, x, y + 1) + 16) >> 5);
	m = clip3(0, 255, (dec_h1(d, x + 1, y) + 16) >> 5);
	j = clip3(0, 255, (dec_tap(dec_b1(d, x, y - 2), dec_b1(d, x, y - 1), dec_b1(d, x, y), dec_b1(d, x, y + 1),
				   dec_b1(d, x, y + 2), dec_b1(d, x, y + 3)) + 512) >> 10);
	if (fx == 2 && fy == 2)
		return j;
	if (fx == 2)
		return fy == 1 ? (b + j + 1) >> 1 : (s + j + 1) >> 1;
	if (fy == 2)
		return fx == 1 ? (h + j + 1) >> 1 : (m + j + 1) >> 1;
	return ((fy == 1 ? b : s) + (fx == 1 ? h : m) + 1) >> 1;
}

static int decode_inter_mb(h264o_dec *d, br *b, dmb *m, int mx, int my, int addr, int skip)
{
	int px, py, cbp = 0, x, y, c;
	int16_t dc16[16];

	if (!d->have_ref)
		return fail(d, "P slice without reference picture");
	memset(m->nnz, 0, sizeof(m->nnz));
	memset(m->i4, 2, sizeof(m->i4));
	pred_mv(d, mx, my, m->slice, addr, skip, &px, &py);
	if (skip) {
		m->type = H264O_MB_PSKIP;
		m->mvx = (int16_t)px;
		m->mvy = (int16_t)py;
	} else {
		unsigned code;

		m->type = H264O_MB_P16x16;
		m->mvx = (int16_t)(px + br_se(b));
		m->mvy = (int16_t)(py + br_se(b));
		code = br_ue(b);
		if (code > 47)
			return fail(d, "bad coded_block_pattern");
		cbp = h264o_cbp_inter_from_code[code];
		if (cbp) {
			int dq = br_se(b);

			if (dq < -26 || dq > 25)
				return fail(d, "mb_qp_delta out of range");
			m->qp = (int8_t)((m->qp + dq + 52) % 52);
		}
	}
	/* luma: quarter-sample interpolation with picture-edge clamping (8.4.2.2.1) */
	for (y = 0; y < 16; y++)
		for (x = 0; x < 16; x++)
			d->cur_y[(my * 16 + y) * d->cw + mx * 16 + x] =
				(uint8_t)dec_luma_sample(d, mx * 16 + x + (m->mvx >> 2), my * 16 + y + (m->mvy >> 2), m->mvx & 3, m->mvy & 3);
	/* chroma: 1/8-sample bilinear (8.4.2.2.2) */
	for (c = 0; c < 2; c++)
		for (y = 0; y < 8; y++)
			for (x = 0; x < 8; x++) {
				int xi = mx * 8 + x + (m->mvx >> 3), yi = my * 8 + y + (m->mvy >> 3);
				int xf = m->mvx & 7, yf = m->mvy & 7;
				int xa = clip3(0, d->cw / 2 - 1, xi), xb = clip3(0, d->cw / 2 - 1, xi + 1);
				int ya = clip3(0, d->ch / 2 - 1, yi), yb = clip3(0, d->ch / 2 - 1, yi + 1);
				int A = d->ref_uv[ya * d->cw + 2 * xa + c], B = d->ref_uv[ya * d->cw + 2 * xb + c];
				int C = d->ref_uv[yb * d->cw + 2 * xa + c], D = d->ref_uv[yb * d->cw + 2 * xb + c];

				d->cur_uv[(my * 8 + y) * d->cw + 2 * (mx * 8 + x) + c] =
					(uint8_t)(((8 - xf) * (8 - yf) * A + xf * (8 - yf) * B + (8 - xf) * yf * C + xf * yf * D + 32) >> 6);
			}
	if (skip)
		return 0;
	return decode_residual_and_recon(d, b, m, mx, my, addr, cbp, 0, dc16);
}

/* ---------------- deblocking (8.7) ---------------- */
static int dec_bs(const dmb *p, int pb, const dmb *q, int qb, int mb_edge)
{
	int pi = p->type <= H264O_MB_I16x16, qi = q->type <= H264O_MB_I16x16;

	if (pi || qi)
		return mb_edge ? 4 : 3;
	if (p->nnz[pb] || q->nnz[qb])
		return 2;
	if (abs(p->mvx - q->mvx) >= 4 || abs(p->mvy - q->mvy) >= 4)
		return 1;
	return 0;
}

static void dec_filter(uint8_t *q0p, int step, int bs, int chroma, int indexA, int indexB)
{
	int alpha = h264o_alpha[indexA], beta = h264o_beta[indexB];
	int p[4], q[4], i;

	for (i = 0; i < (chroma ? 2 : 4); i++) {
		p[i] = q0p[-(i + 1) * step];
		q[i] = q0p[i * step];
	}
	if (!(abs(p[0] - q[0]) < alpha && abs(p[1] - p[0]) < beta && abs(q[1] - q[0]) < beta))
		return;
	if (bs < 4) {
		int tc0 = h264o_tc0[indexA][bs - 1], tc, delta;
		int ap = chroma ? 0 : abs(p[2] - p[0]), aq = chroma ? 0 : abs(q[2] - q[0]);

		tc = chroma ? tc0 + 1 : tc0 + (ap < beta) + (aq < beta);
		delta = clip3(-tc, tc, ((((q[0] - p[0]) * 4) + (p[1] - q[1]) + 4) >> 3));
		q0p[-step] = (uint8_t)clip8(p[0] + delta);
		q0p[0] = (uint8_t)clip8(q[0] - delta);
		if (!chroma && ap < beta)
			q0p[-2 * step] = (uint8_t)(p[1] + clip3(-tc0, tc0, (p[2] + ((p[0] + q[0] + 1) >> 1) - (p[1] << 1)) >> 1));
		if (!chroma && aq < beta)
			q0p[step] = (uint8_t)(q[1] + clip3(-tc0, tc0, (q[2] + ((p[0] + q[0] + 1) >> 1) - (q[1] << 1)) >> 1));
	} else {
		int ap = chroma ? 0 : abs(p[2] - p[0]), aq = chroma ? 0 : abs(q[2] - q[0]);
		int strong = abs(p[0] - q[0]) < ((alpha >> 2) + 2);

		if (!chroma && ap < beta && strong) {
			q0p[-step] = (uint8_t)((p[2] + 2 * p[1] + 2 * p[0] + 2 * q[0] + q[1] + 4) >> 3);
			q0p[-2 * step] = (uint8_t)((p[2] + p[1] + p[0] + q[0] + 2) >> 2);
			q0p[-3 * step] = (uint8_t)((2 * p[3] + 3 * p[2] + p[1] + p[0] + q[0] + 4) >> 3);
		} else {
			q0p[-step] = (uint8_t)((2 * p[1] + p[0] + q[1] + 2) >> 2);
		}
		if (!chroma && aq < beta && strong) {
			q0p[0] = (uint8_t)((p[1] + 2 * p[0] + 2 * q[0] + 2 * q[1] + q[2] + 4) >> 3);
			q0p[step] = (uint8_t)((p[0] + q[0] + q[1] + q[2] + 2) >> 2);
			q0p[2 * step] = (uint8_t)((2 * q[3] + 3 * q[2] + q[1] + q[0] + p[0] + 4) >> 3);
		} else {
			q0p[0] = (uint8_t)((2 * q[1] + q[0] + p[1] + 2) >> 2);
		}
	}
}

static void deblock_picture(h264o_dec *d)
{
	int mx, my, e, k, c;

	for (my = 0; my < d->mb_h; my++)
		for (mx = 0; mx < d->mb_w; mx++) {
			const dmb *q = &d->mbs[my * d->mb_w + mx];
			int idc = d->slice_idc[q->slice], oa = d->slice_a[q->slice], ob = d->slice_b[q->slice];
			int dir;

			if (idc == 1)
				continue;
			for (dir = 0; dir < 2; dir++) /* 0: vertical edges, 1: horizontal edges */
				for (e = 0; e < 4; e++) {
					const dmb *p = q;
					int qpav, qpcav, iA, iB, iAc, iBc;

					if (e == 0) {
						int nx = dir ? mx : mx - 1, ny = dir ? my - 1 : my;

						if (nx < 0 || ny < 0)
							continue;
						p = &d->mbs[ny * d->mb_w + nx];
						if (idc == 2 && p->slice != q->slice)
							continue;
					}
					{
						/* 8.7.2.2: qPp of an I_PCM macroblock is 0 */
						const int pq = p->type == H264O_MB_IPCM ? 0 : p->qp, qq = q->type == H264O_MB_IPCM ? 0 : q->qp;

						qpav = (pq + qq + 1) >> 1;
						qpcav = (chroma_qp(d, pq) + chroma_qp(d, qq) + 1) >> 1;
					}
					iA = clip3(0, 51, qpav + oa);
					iB = clip3(0, 51, qpav + ob);
					iAc = clip3(0, 51, qpcav + oa);
					iBc = clip3(0, 51, qpcav + ob);
					for (k = 0; k < 16; k++) {
						int blk = k >> 2, pb, qb, bs;
						uint8_t *pix;

						if (dir == 0) {
							qb = e + 4 * blk;
							pb = e == 0 ? 3 + 4 * blk : e - 1 + 4 * blk;
							pix = d->cur_y + (my * 16 + k) * d->cw + mx * 16 + 4 * e;
						} else {
							qb = blk + 4 * e;
							pb = e == 0 ? blk + 12 : blk + 4 * (e - 1);
							pix = d->cur_y + (my * 16 + 4 * e) * d->cw + mx * 16 + k;
						}
						bs = dec_bs(p, pb, q, qb, e == 0);
						if (bs)
							dec_filter(pix, dir ? d->cw : 1, bs, 0, iA, iB);
						/* chroma: luma edges 0 and 2, chroma sample k>>1 of the
						 * same luma-edge segment; visit once per chroma sample */
						if (!(e & 1) && !(k & 1) && bs)
							for (c = 0; c < 2; c++) {
								uint8_t *cp;

								if (dir == 0)
									cp = d->cur_uv + (my * 8 + (k >> 1)) * d->cw + 2 * (mx * 8 + 2 * e) + c;
								else
									cp = d->cur_uv + (my * 8 + 2 * e) * d->cw + 2 * (mx * 8 + (k >> 1)) + c;
								dec_filter(cp, dir ? d->cw : 2, bs, 1, iAc, iBc);
							}
					}
				}
		}
}

/* ---------------- slices and NALs ---------------- */
static int decode_slice(h264o_dec *d, br *b, int nal_type, int nal_ref_idc)
{
	int first_mb, stype, addr, idc = 0, oa = 0, ob = 0, qp;
	int skip_run = -1;

	(void)nal_ref_idc;
	if (!d->have_sps || !d->have_pps)
		return fail(d, "slice before parameter sets");
	first_mb = (int)br_ue(b);
	stype = (int)br_ue(b) % 5;
	if (stype != 0 && stype != 2)
		return fail(d, "only I and P slices supported");
	br_ue(b); /* pps id */
	br_u(b, d->log2_max_frame_num);
	if (nal_type == 5)
		br_ue(b); /* idr_pic_id */
	if (d->poc_type == 0)
		return fail(d, "poc type 0 unsupported in slice header");
	if (stype == 0) {
		if (br_u(b, 1))
			return fail(d, "num_ref_idx override unsupported");
		if (br_u(b, 1))
			return fail(d, "ref pic list modification unsupported");
	}
	if (nal_ref_idc) {
		if (nal_type == 5) {
			br_u(b, 1);
			br_u(b, 1);
		} else if (br_u(b, 1)) {
			return fail(d, "adaptive ref pic marking unsupported");
		}
	}
	qp = d->init_qp + br_se(b);
	if (d->deblock_ctrl) {
		idc = (int)br_ue(b);
		if (idc != 1) {
			oa = 2 * br_se(b);
			ob = 2 * br_se(b);
		}
	}
	if (b->err)
		return fail(d, "slice header truncated");
	if (first_mb == 0) {
		if (d->mbs_done != 0 && d->mbs_done != d->mb_w * d->mb_h)
			return fail(d, "new picture before the previous one completed");
		d->mbs_done = 0;
		d->n_slices = 0;
	}
	if (first_mb != d->mbs_done)
		return fail(d, "first_mb_in_slice does not continue the picture");
	d->slice_idc[d->n_slices] = idc;
	d->slice_a[d->n_slices] = oa;
	d->slice_b[d->n_slices] = ob;
	addr = first_mb;
	for (;;) {
		dmb *m;
		int mx, my, r;

		if (addr >= d->mb_w * d->mb_h)
			return fail(d, "too many macroblocks in slice");
		m = &d->mbs[addr];
		mx = addr % d->mb_w;
		my = addr / d->mb_w;
		m->slice = (int8_t)d->n_slices;
		m->qp = (int8_t)qp;
		if (stype == 0) {
			if (skip_run < 0)
				skip_run = (int)br_ue(b);
			if (skip_run > 0) {
				skip_run--;
				r = decode_inter_mb(d, b, m, mx, my, addr, 1);
				if (r < 0)
					return r;
				addr++;
				if (skip_run == 0 && !br_more(b))
					break;
				continue;
			}
			skip_run = -1;
			if (!br_more(b))
				break;
			{
				unsigned t = br_ue(b);

				if (t == 0)
					r = decode_inter_mb(d, b, m, mx, my, addr, 0);
				else if (t >= 5)
					r = decode_intra_mb(d, b, m, mx, my, addr, (int)t - 5);
				else
					return fail(d, "P partition type unsupported");
			}
		} else {
			r = decode_intra_mb(d, b, m, mx, my, addr, (int)br_ue(b));
		}
		if (r < 0)
			return r;
		qp = m->qp;
		addr++;
		if (b->err || b->pos > b->nbits) /* (the reader lets the stop bit itself be read: no syntax element may hold it) */
			return fail(d, "slice data truncated");
		if (!br_more(b))
			break;
	}
	d->mbs_done = addr;
	d->n_slices++;
	if (d->mbs_done == d->mb_w * d->mb_h) {
		uint8_t *t;

		deblock_picture(d);
		t = d->cur_y; d->cur_y = d->ref_y; d->ref_y = t;
		t = d->cur_uv; d->cur_uv = d->ref_uv; d->ref_uv = t;
		d->have_ref = 1;
		return 1;
	}
	return 0;
}

int h264o_dec_decode(h264o_dec *d, const uint8_t *data, int size)
{
	int i = 0, pictures = 0;

	d->err[0] = 0;
	if (d->rbsp_cap < size + 8) {
		free(d->rbsp);
		d->rbsp_cap = size + 8;
		d->rbsp = malloc((size_t)d->rbsp_cap);
		if (!d->rbsp)
			return fail(d, "out of memory");
	}
	while (i + 3 < size) {
		int start, end, n = 0, k, zeros = 0, hdr, last;
		br b;

		/* find start code 00 00 01 */
		if (!(data[i] == 0 && data[i + 1] == 0 && data[i + 2] == 1)) {
			if (data[i] != 0) /* Annex B: only zero bytes lie between NAL units (leading_zero_8bits, trailing_zero_8bits) */
				return fail(d, "bytes that belong to no NAL unit");
			i++;
			continue;
		}
		start = i + 3;
		end = start;
		while (end + 2 < size && !(data[end] == 0 && data[end + 1] == 0 && (data[end + 2] == 1 || data[end + 2] == 0)))
			end++;
		if (end + 2 >= size)
			end = size;
		if (start >= end)
			return fail(d, "empty NAL");
		hdr = data[start];
		for (k = start + 1; k < end; k++) {
			if (zeros >= 2 && data[k] == 3) {
				zeros = 0;
				continue;
			}
			if (zeros >= 2 && data[k] < 3)
				return fail(d, "start code emulation inside NAL");
			d->rbsp[n++] = data[k];
			zeros = data[k] == 0 ? zeros + 1 : 0;
		}
		/* locate rbsp stop bit */
		last = n - 1;
		while (last >= 0 && d->rbsp[last] == 0)
			last--;
		if (last < 0)
			return fail(d, "NAL without stop bit");
		b.p = d->rbsp;
		b.pos = 0;
		b.err = 0;
		{
			int bit = 0;

			while (!((d->rbsp[last] >> bit) & 1))
				bit++;
			b.nbits = last * 8 + (7 - bit);
		}
		switch (hdr & 0x1f) {
		case 7:
			if (parse_sps(d, &b) < 0)
				return -1;
			break;
		case 8:
			if (parse_pps(d, &b) < 0)
				return -1;
			break;
		case 1:
		case 5: {
			int r = decode_slice(d, &b, hdr & 0x1f, (hdr >> 5) & 3);

			if (r < 0)
				return -1;
			pictures += r;
			break;
		}
		case 9: /* access unit delimiter: counted, content ignored */
			d->n_aud++;
			break;
		default:
			break; /* ignore other NAL types */
		}
		i = end;
	}
	return pictures;
}
