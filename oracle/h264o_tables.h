/*
 * oracle/h264o_tables.h -- ITU-T H.264 constant tables for the scalar oracle.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under oracle/ is linked, imported or
 * executed by the product path (cedarx_h264_encoder_amd/); only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * PARITY UNPINNED at macroblock level: the reference (libv/cedarx_h264_encoder)
 * has no macroblock software (kernel/cedar.c:1176 triggers fixed-function
 * silicon), so these tables restate the H.264 standard, not the reference.
 * They are written in (length, code) form; tests/test_tables.py checks every
 * VLC table is prefix-free (Kraft sum) and cross-checks the product's
 * independently written bit-string tables against these.
 */
#ifndef H264O_TABLES_H
#define H264O_TABLES_H

#include <stdint.h>

/* 4x4 zig-zag scan (frame): scan index -> raster index (row*4+col). [spec 8.5.6] */
static const uint8_t h264o_zigzag4x4[16] = {
	0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15
};

/* luma 4x4 block coding order -> (x,y) in 4x4-block units. [spec 6.4.3] */
static const uint8_t h264o_blk_x[16] = { 0, 1, 0, 1, 2, 3, 2, 3, 0, 1, 0, 1, 2, 3, 2, 3 };
static const uint8_t h264o_blk_y[16] = { 0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3 };

/* Forward quantiser multipliers MF[qp%6][class], class 0 = positions with
 * both coordinates even, 1 = both odd, 2 = mixed. [JM / spec-derived] */
static const uint16_t h264o_quant_mf[6][3] = {
	{ 13107, 5243, 8066 }, { 11916, 4660, 7490 }, { 10082, 4194, 6554 },
	{  9362, 3647, 5825 }, {  8192, 3355, 5243 }, {  7282, 2893, 4559 },
};

/* Dequantiser scale V[qp%6][class] = normAdjust4x4. [spec 8.5.9] */
static const uint8_t h264o_dequant_v[6][3] = {
	{ 10, 16, 13 }, { 11, 18, 14 }, { 13, 20, 16 },
	{ 14, 23, 18 }, { 16, 25, 20 }, { 18, 29, 23 },
};

/* class of raster position i (0..15) */
static const uint8_t h264o_pos_class[16] = {
	0, 2, 0, 2,
	2, 1, 2, 1,
	0, 2, 0, 2,
	2, 1, 2, 1
};

/* QPc as a function of qPI. [spec table 8-15] */
static const uint8_t h264o_chroma_qp[52] = {
	 0,  1,  2,  3,  4,  5,  6,  7,  8,  9, 10, 11, 12, 13, 14, 15,
	16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30,
	31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38,
	39, 39, 39, 39
};

/* Deblocking alpha / beta / tc0. [spec tables 8-16, 8-17] */
static const uint8_t h264o_alpha[52] = {
	0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
	4, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22, 25, 28,
	32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182,
	203, 226, 255, 255
};
static const uint8_t h264o_beta[52] = {
	0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
	2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7, 8, 8,
	9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16,
	17, 17, 18, 18
};
static const uint8_t h264o_tc0[52][3] = {
	{ 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 },
	{ 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 },
	{ 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 },
	{ 0, 0, 1 }, { 0, 0, 1 }, { 0, 0, 1 }, { 0, 0, 1 },
	{ 0, 1, 1 }, { 0, 1, 1 },
	{ 1, 1, 1 }, { 1, 1, 1 }, { 1, 1, 1 }, { 1, 1, 1 },
	{ 1, 1, 2 }, { 1, 1, 2 }, { 1, 1, 2 }, { 1, 1, 2 },
	{ 1, 2, 3 }, { 1, 2, 3 },
	{ 2, 2, 3 }, { 2, 2, 4 }, { 2, 3, 4 }, { 2, 3, 4 },
	{ 3, 3, 5 }, { 3, 4, 6 }, { 3, 4, 6 },
	{ 4, 5, 7 }, { 4, 5, 8 }, { 4, 6, 9 },
	{ 5, 7, 10 }, { 6, 8, 11 }, { 6, 8, 13 },
	{ 7, 10, 14 }, { 8, 11, 16 }, { 9, 12, 18 },
	{ 10, 13, 20 }, { 11, 15, 23 }, { 13, 17, 25 }
};

/* coded_block_pattern me(v) mapping, ChromaArrayType 1: codeNum -> cbp.
 * [spec table 9-4]. Encoder uses the inverse (built at init). */
static const uint8_t h264o_cbp_intra_from_code[48] = {
	47, 31, 15, 0, 23, 27, 29, 30, 7, 11, 13, 14, 39, 43, 45, 46,
	16, 3, 5, 10, 12, 19, 21, 26, 28, 35, 37, 42, 44, 1, 2, 4,
	8, 17, 18, 20, 24, 6, 9, 22, 25, 32, 33, 34, 36, 40, 38, 41
};
static const uint8_t h264o_cbp_inter_from_code[48] = {
	0, 16, 1, 2, 4, 8, 32, 3, 5, 10, 12, 15, 47, 7, 11, 13,
	14, 6, 9, 31, 35, 37, 42, 44, 33, 34, 36, 40, 39, 43, 45, 46,
	17, 18, 20, 24, 19, 21, 26, 28, 23, 27, 29, 30, 22, 25, 38, 41
};

/* coeff_token: [table][4*total_coeff + trailing_ones]; table 0: 0<=nC<2,
 * 1: 2<=nC<4, 2: 4<=nC<8, 3: 8<=nC (6-bit FLC). len 0 = invalid. [spec table 9-5] */
static const uint8_t h264o_coeff_token_len[4][68] = {
	{
		1, 0, 0, 0,
		6, 2, 0, 0,     8, 6, 3, 0,     9, 8, 7, 5,    10, 9, 8, 6,
		11, 10, 9, 7,   13, 11, 10, 8,  13, 13, 11, 9,  13, 13, 13, 10,
		14, 14, 13, 11, 14, 14, 14, 13, 15, 15, 14, 14, 15, 15, 15, 14,
		16, 15, 15, 15, 16, 16, 16, 15, 16, 16, 16, 16, 16, 16, 16, 16,
	},
	{
		2, 0, 0, 0,
		6, 2, 0, 0,     6, 5, 3, 0,     7, 6, 6, 4,     8, 6, 6, 4,
		8, 7, 7, 5,     9, 8, 8, 6,    11, 9, 9, 6,    11, 11, 11, 7,
		12, 11, 11, 9,  12, 12, 12, 11, 12, 12, 12, 11, 13, 13, 13, 12,
		13, 13, 13, 13, 13, 14, 13, 13, 14, 14, 14, 13, 14, 14, 14, 14,
	},
	{
		4, 0, 0, 0,
		6, 4, 0, 0,     6, 5, 4, 0,     6, 5, 5, 4,     7, 5, 5, 4,
		7, 5, 5, 4,     7, 6, 6, 4,     7, 6, 6, 4,     8, 7, 7, 5,
		8, 8, 7, 6,     9, 8, 8, 7,     9, 9, 8, 8,     9, 9, 9, 8,
		10, 9, 9, 9,    10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10,
	},
	{
		6, 0, 0, 0,
		6, 6, 0, 0,     6, 6, 6, 0,     6, 6, 6, 6,     6, 6, 6, 6,
		6, 6, 6, 6,     6, 6, 6, 6,     6, 6, 6, 6,     6, 6, 6, 6,
		6, 6, 6, 6,     6, 6, 6, 6,     6, 6, 6, 6,     6, 6, 6, 6,
		6, 6, 6, 6,     6, 6, 6, 6,     6, 6, 6, 6,     6, 6, 6, 6,
	}
};
static const uint8_t h264o_coeff_token_bits[4][68] = {
	{
		1, 0, 0, 0,
		5, 1, 0, 0,     7, 4, 1, 0,     7, 6, 5, 3,     7, 6, 5, 3,
		7, 6, 5, 4,     15, 6, 5, 4,    11, 14, 5, 4,   8, 10, 13, 4,
		15, 14, 9, 4,   11, 10, 13, 12, 15, 14, 9, 12,  11, 10, 13, 8,
		15, 1, 9, 12,   11, 14, 13, 8,  7, 10, 9, 12,   4, 6, 5, 8,
	},
	{
		3, 0, 0, 0,
		11, 2, 0, 0,    7, 7, 3, 0,     7, 10, 9, 5,    7, 6, 5, 4,
		4, 6, 5, 6,     7, 6, 5, 8,     15, 6, 5, 4,    11, 14, 13, 4,
		15, 10, 9, 4,   11, 14, 13, 12, 8, 10, 9, 8,    15, 14, 13, 12,
		11, 10, 9, 12,  7, 11, 6, 8,    9, 8, 10, 1,    7, 6, 5, 4,
	},
	{
		15, 0, 0, 0,
		15, 14, 0, 0,   11, 15, 13, 0,  8, 12, 14, 12,  15, 10, 11, 11,
		11, 8, 9, 10,   9, 14, 13, 9,   8, 10, 9, 8,    15, 14, 13, 13,
		11, 14, 10, 12, 15, 10, 13, 12, 11, 14, 9, 12,  8, 10, 13, 8,
		13, 7, 9, 12,   9, 12, 11, 10,  5, 8, 7, 6,     1, 4, 3, 2,
	},
	{
		3, 0, 0, 0,
		0, 1, 0, 0,     4, 5, 6, 0,     8, 9, 10, 11,   12, 13, 14, 15,
		16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31,
		32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47,
		48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63,
	}
};

/* chroma DC coeff_token (nC == -1): [4*total_coeff + trailing_ones]. */
static const uint8_t h264o_chroma_dc_coeff_token_len[20] = {
	2, 0, 0, 0,
	6, 1, 0, 0,
	6, 6, 3, 0,
	6, 7, 7, 6,
	6, 8, 8, 7,
};
static const uint8_t h264o_chroma_dc_coeff_token_bits[20] = {
	1, 0, 0, 0,
	7, 1, 0, 0,
	4, 6, 1, 0,
	3, 3, 2, 5,
	2, 3, 2, 0,
};

/* total_zeros for 4x4 blocks: [total_coeff-1][total_zeros]. [spec tables 9-7, 9-8] */
static const uint8_t h264o_total_zeros_len[15][16] = {
	{ 1, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 9 },
	{ 3, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 6, 6, 6, 6 },
	{ 4, 3, 3, 3, 4, 4, 3, 3, 4, 5, 5, 6, 5, 6 },
	{ 5, 3, 4, 4, 3, 3, 3, 4, 3, 4, 5, 5, 5 },
	{ 4, 4, 4, 3, 3, 3, 3, 3, 4, 5, 4, 5 },
	{ 6, 5, 3, 3, 3, 3, 3, 3, 4, 3, 6 },
	{ 6, 5, 3, 3, 3, 2, 3, 4, 3, 6 },
	{ 6, 4, 5, 3, 2, 2, 3, 3, 6 },
	{ 6, 6, 4, 2, 2, 3, 2, 5 },
	{ 5, 5, 3, 2, 2, 2, 4 },
	{ 4, 4, 3, 3, 1, 3 },
	{ 4, 4, 2, 1, 3 },
	{ 3, 3, 1, 2 },
	{ 2, 2, 1 },
	{ 1, 1 },
};
static const uint8_t h264o_total_zeros_bits[15][16] = {
	{ 1, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 1 },
	{ 7, 6, 5, 4, 3, 5, 4, 3, 2, 3, 2, 3, 2, 1, 0 },
	{ 5, 7, 6, 5, 4, 3, 4, 3, 2, 3, 2, 1, 1, 0 },
	{ 3, 7, 5, 4, 6, 5, 4, 3, 3, 2, 2, 1, 0 },
	{ 5, 4, 3, 7, 6, 5, 4, 3, 2, 1, 1, 0 },
	{ 1, 1, 7, 6, 5, 4, 3, 2, 1, 1, 0 },
	{ 1, 1, 5, 4, 3, 3, 2, 1, 1, 0 },
	{ 1, 1, 1, 3, 3, 2, 2, 1, 0 },
	{ 1, 0, 1, 3, 2, 1, 1, 1 },
	{ 1, 0, 1, 3, 2, 1, 1 },
	{ 0, 1, 1, 2, 1, 3 },
	{ 0, 1, 1, 1, 1 },
	{ 0, 1, 1, 1 },
	{ 0, 1, 1 },
	{ 0, 1 },
};

/* total_zeros for chroma DC 2x2: [total_coeff-1][total_zeros]. [spec table 9-9a] */
static const uint8_t h264o_chroma_dc_total_zeros_len[3][4] = {
	{ 1, 2, 3, 3 }, { 1, 2, 2, 0 }, { 1, 1, 0, 0 }
};
static const uint8_t h264o_chroma_dc_total_zeros_bits[3][4] = {
	{ 1, 1, 1, 0 }, { 1, 1, 0, 0 }, { 1, 0, 0, 0 }
};

/* run_before: [min(zeros_left,7)-1][run_before]. [spec table 9-10] */
static const uint8_t h264o_run_before_len[7][15] = {
	{ 1, 1 },
	{ 1, 2, 2 },
	{ 2, 2, 2, 2 },
	{ 2, 2, 2, 3, 3 },
	{ 2, 2, 3, 3, 3, 3 },
	{ 2, 3, 3, 3, 3, 3, 3 },
	{ 3, 3, 3, 3, 3, 3, 3, 4, 5, 6, 7, 8, 9, 10, 11 },
};
static const uint8_t h264o_run_before_bits[7][15] = {
	{ 1, 0 },
	{ 1, 1, 0 },
	{ 3, 2, 1, 0 },
	{ 3, 2, 1, 1, 0 },
	{ 3, 2, 3, 2, 1, 0 },
	{ 3, 0, 1, 3, 2, 5, 4 },
	{ 7, 6, 5, 4, 3, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1 },
};

/* Encoder-side (non-normative) Lagrangian weight per QP, used by the motion
 * search and the I4x4-vs-I16x16 decision: round(2^((qp-12)/6)), min 1.
 * Design choice of this build (the reference's silicon policy is unknown). */
static const uint8_t h264o_lambda[52] = {
	1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
	2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6, 6, 7, 8, 9,
	10, 11, 13, 14, 16, 18, 20, 23, 25, 29, 32, 36, 40, 45, 51, 57,
	64, 72, 81, 91
};

#endif /* H264O_TABLES_H */
